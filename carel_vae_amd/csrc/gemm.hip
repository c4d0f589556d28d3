// bf16 MFMA GEMM for the encoder linears of CAREL-VAE (replaces the nn.Linear calls inside HF
// BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput, reached from
// drl_classifier_ec_mmd_final_mul.py:202-206, and their autograd backward, :841).
//
//   C[M,N] = op(A) * op(B)   fp32 accumulate on v_mfma_f32_16x16x32_bf16
//   forward  (NT): A = activations [M,K] row-major, B = nn.Linear weight [N,K] row-major
//   dgrad    (NN): A = dY [M,K=Nout],   B = weight [K=Nout, N=Nin] row-major  (COL image + tr reads)
//   wgrad    (TN): A = dY [K=T, M=Nout], B = X [K=T, N=Nin]; both COL images; split-K over T into slabs
//
// Block tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles.
// Operands are staged global->LDS with global_load_lds_dwordx4 (double buffered, 64 KiB LDS,
// 2 blocks/CU); images are XOR-swizzled (carel_common.h) so fragment reads are bank-conflict free.
// The MFMA is issued with the operands swapped (a = B fragment, b = A fragment) so that each lane's 4
// accumulator registers are 4 CONSECUTIVE columns of one C row -> 8/16-byte epilogue accesses.
#include "carel_common.h"
#include "carel_hip_internal.h"

namespace carel {

enum : int {
  EPI_BIAS_BF16 = 0,        // out0(bf16) = acc + bias?            (QKV, dgrad of out-proj)
  EPI_BIAS_GELU = 1,        // out0(bf16) = u = acc+bias ; out1(bf16) = gelu(u)       (FFN1)
  EPI_BIAS_DROP_RESID = 2,  // outf(f32) = dropout(acc + bias) + resid(f32)           (out-proj, FFN2)
  EPI_DGELU_BF16 = 3,       // out0(bf16) = acc * gelu'(aux_bf16)                     (dgrad of FFN2)
  EPI_ADD_F32 = 4,          // outf(f32) = acc + resid(f32)?                          (dgrad of QKV / FFN1)
  EPI_SLAB_F32 = 5,         // outf[z](f32) = acc                                     (wgrad split-K)
};

struct GemmParams {
  const bf16_t* A; const bf16_t* B;
  long lda, ldb;
  int M, N, K;              // K = contraction length handled by ONE z-slice
  bf16_t* out0; bf16_t* out1; float* outf;
  long ldc;
  const float* bias;        // [N] or null
  const float* resid;       // [M,ldc] f32 or null
  const bf16_t* aux;        // [M,ldc] bf16 (pre-GELU) for EPI_DGELU
  Dropout drop;
  int tiles_m, tiles_n;
};

template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware, bijective tile remap: blocks that share an XCD (bid % 8) get neighbouring tiles
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int tid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tm = tid / p.tiles_n, tn = tid - tm * p.tiles_n;
  const long m0 = (long)tm * 128, n0 = (long)tn * 128;
  const long kbase = (long)blockIdx.z * p.K;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int kt, int buf) {
    char* ta = smem + buf * 32768;
    char* tb = ta + 16384;
    const long k0 = kbase + (long)kt * 64;
    if (AT) stage_col_image(p.A, p.lda, k0, m0, ta); else stage_row_image<128>(p.A, p.lda, m0, k0, ta);
    if (BT) stage_col_image(p.B, p.ldb, k0, n0, tb); else stage_row_image<128>(p.B, p.ldb, n0, k0, tb);
  };

  const int nk = p.K >> 6;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* ta = smem + (kt & 1) * 32768;
    const char* tb = ta + 16384;
#pragma unroll
    for (int ks = 0; ks < 64; ks += 32) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = AT ? frag16_col(ta, wr * 64 + i * 16, ks) : frag16_row(ta, wr * 64 + i * 16, ks);
        fb[i] = BT ? frag16_col(tb, wc * 64 + i * 16, ks) : frag16_row(tb, wc * 64 + i * 16, ks);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);   // swapped: D[n][m]
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  // lane: C row = m0 + wr*64 + i*16 + (lane&15); columns n0 + wc*64 + j*16 + (lane>>4)*4 + [0..3]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long row = m0 + wr * 64 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long col = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
      f32x4 v = acc[i][j];
      const long off = row * p.ldc + col;
      if (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_DROP_RESID) {
        if (p.bias) {
          const float4 b = *(const float4*)(p.bias + col);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
      }
      if (EPI == EPI_BIAS_BF16) {
        uint2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *(uint2*)(p.out0 + off) = o;
      } else if (EPI == EPI_BIAS_GELU) {
        uint2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *(uint2*)(p.out0 + off) = o;
        // GELU is evaluated on the bf16-rounded pre-activation that backward will read back
        float u0 = bf2f(f2bf(v[0])), u1 = bf2f(f2bf(v[1])), u2 = bf2f(f2bf(v[2])), u3 = bf2f(f2bf(v[3]));
        uint2 g = {pack2bf(gelu_erf(u0), gelu_erf(u1)), pack2bf(gelu_erf(u2), gelu_erf(u3))};
        *(uint2*)(p.out1 + off) = g;
      } else if (EPI == EPI_BIAS_DROP_RESID) {
        const float4 r = *(const float4*)(p.resid + off);
        const uint32_t e = (uint32_t)off;
        float4 o;
        o.x = v[0] * dropout_mult(p.drop, e + 0) + r.x;
        o.y = v[1] * dropout_mult(p.drop, e + 1) + r.y;
        o.z = v[2] * dropout_mult(p.drop, e + 2) + r.z;
        o.w = v[3] * dropout_mult(p.drop, e + 3) + r.w;
        *(float4*)(p.outf + off) = o;
      } else if (EPI == EPI_DGELU_BF16) {
        const uint2 a = *(const uint2*)(p.aux + off);
        const float u0 = bf2f((bf16_t)(a.x & 0xffff)), u1 = bf2f((bf16_t)(a.x >> 16));
        const float u2 = bf2f((bf16_t)(a.y & 0xffff)), u3 = bf2f((bf16_t)(a.y >> 16));
        uint2 o = {pack2bf(v[0] * gelu_erf_grad(u0), v[1] * gelu_erf_grad(u1)),
                   pack2bf(v[2] * gelu_erf_grad(u2), v[3] * gelu_erf_grad(u3))};
        *(uint2*)(p.out0 + off) = o;
      } else if (EPI == EPI_ADD_F32) {
        float4 o = {v[0], v[1], v[2], v[3]};
        if (p.resid) {
          const float4 r = *(const float4*)(p.resid + off);
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        *(float4*)(p.outf + off) = o;
      } else {  // EPI_SLAB_F32
        float4 o = {v[0], v[1], v[2], v[3]};
        *(float4*)(p.outf + (long)blockIdx.z * p.M * p.ldc + off) = o;
      }
    }
  }
}

template <bool AT, bool BT, int EPI>
static int launch(const GemmParams& p, int splits, hipStream_t s) {
  dim3 grid(p.tiles_m * p.tiles_n, 1, splits);
  hipLaunchKernelGGL((gemm_kernel<AT, BT, EPI>), grid, dim3(256), 0, s, p);
  return check_launch("gemm_kernel");
}

// Sum `splits` fp32 slabs [splits][n] -> out[n] (+ optional accumulate into out)
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, long n, int splits,
                                   int accumulate) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    float4 a = *(const float4*)(slabs + i);
    for (int z = 1; z < splits; ++z) {
      const float4 b = *(const float4*)(slabs + (long)z * n + i);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (accumulate) {
      const float4 o = *(const float4*)(out + i);
      a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    }
    *(float4*)(out + i) = a;
  }
}

}  // namespace carel

using namespace carel;

// ---- optional per-launch HIP-event timing of the GEMM kernel (bench.py's roofline leg) -------------
#include <vector>
namespace {
struct GemmProf {
  bool on = false;
  std::vector<hipEvent_t> ev;      // start/stop pairs
  std::vector<double> flops;
  size_t used = 0;
} g_prof;
struct ProfScope {
  hipStream_t s; bool active;
  ProfScope(hipStream_t st, double fl) : s(st), active(false) {
    if (!g_prof.on || g_prof.used + 2 > g_prof.ev.size()) return;
    active = true;
    g_prof.flops.push_back(fl);
    (void)hipEventRecord(g_prof.ev[g_prof.used], s);
  }
  ~ProfScope() {
    if (!active) return;
    (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s);
    g_prof.used += 2;
  }
};
}  // namespace

extern "C" int carel_profile_gemm(int enable, int max_launches) {
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear(); g_prof.flops.clear(); g_prof.used = 0; g_prof.on = false;
  if (!enable) return CAREL_OK;
  if (max_launches < 1) return set_error(CAREL_ERR_ARG, "carel_profile_gemm: max_launches must be positive");
  g_prof.ev.resize((size_t)max_launches * 2);
  for (auto& e : g_prof.ev)
    if (hipEventCreate(&e) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_profile_gemm: hipEventCreate failed");
  g_prof.on = true;
  return CAREL_OK;
}

// Host-synchronising read-out: sums the recorded launches (ms, algorithmic flops 2*M*N*K, count).
extern "C" int carel_profile_gemm_read(double* total_ms, double* total_flops, int64_t* launches) {
  if (!total_ms || !total_flops || !launches) return set_error(CAREL_ERR_ARG, "carel_profile_gemm_read: null output");
  double ms = 0.0, fl = 0.0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_profile_gemm_read: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_profile_gemm_read: elapsed failed");
    ms += t; fl += g_prof.flops[i / 2];
  }
  *total_ms = ms; *total_flops = fl; *launches = (int64_t)(g_prof.used / 2);
  g_prof.used = 0; g_prof.flops.clear();
  return CAREL_OK;
}

static int gemm_shape_ok(int M, int N, int K, int splits) {
  if (M <= 0 || N <= 0 || K <= 0 || splits <= 0) return 0;
  if (M % 128 || N % 128) return 0;
  if (K % (64 * splits)) return 0;
  return 1;
}

extern "C" int carel_gemm_bf16(const carel_gemm_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a) return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: null args");
  const int splits = a->splits > 0 ? a->splits : 1;
  if (!gemm_shape_ok(a->M, a->N, a->K, splits))
    return set_error(CAREL_ERR_SHAPE, "carel_gemm_bf16: M,N must be multiples of 128 and K of 64*splits (M=%d N=%d K=%d splits=%d)",
                     a->M, a->N, a->K, splits);
  if (!a->A || !a->B) return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: null operand");
  if (((uintptr_t)a->A | (uintptr_t)a->B) & 15 || (a->lda & 7) || (a->ldb & 7) || (a->ldc & 3))
    return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: operands must be 16-byte aligned, ld multiples of 8");
  GemmParams p;
  p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.lda = a->lda; p.ldb = a->ldb;
  p.M = a->M; p.N = a->N; p.K = a->K / splits;
  p.out0 = (bf16_t*)a->out_bf16; p.out1 = (bf16_t*)a->out2_bf16; p.outf = (float*)a->out_f32; p.ldc = a->ldc;
  p.bias = (const float*)a->bias; p.resid = (const float*)a->resid_f32; p.aux = (const bf16_t*)a->aux_bf16;
  p.drop = make_dropout(a->drop_seed, a->drop_site, a->drop_p, a->drop_idx_offset);
  p.tiles_m = a->M / 128; p.tiles_n = a->N / 128;
  const int form = a->form, epi = a->epilogue;
  if (splits != 1 && epi != EPI_SLAB_F32) return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: split-K only with the slab epilogue");
#define NEED(ptr, what) if (!(ptr)) return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: epilogue needs " what)
  switch (epi) {
    case EPI_BIAS_BF16: NEED(p.out0, "out_bf16"); break;
    case EPI_BIAS_GELU: NEED(p.out0, "out_bf16"); NEED(p.out1, "out2_bf16"); break;
    case EPI_BIAS_DROP_RESID: NEED(p.outf, "out_f32"); NEED(p.resid, "resid_f32"); break;
    case EPI_DGELU_BF16: NEED(p.out0, "out_bf16"); NEED(p.aux, "aux_bf16"); break;
    case EPI_ADD_F32: NEED(p.outf, "out_f32"); break;
    case EPI_SLAB_F32: NEED(p.outf, "out_f32"); break;
    default: return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: unknown epilogue %d", epi);
  }
#undef NEED
  ProfScope prof_scope(stream, 2.0 * (double)a->M * (double)a->N * (double)a->K);
  if (form == CAREL_GEMM_NT) {
    switch (epi) {
      case EPI_BIAS_BF16: return launch<false, false, EPI_BIAS_BF16>(p, 1, stream);
      case EPI_BIAS_GELU: return launch<false, false, EPI_BIAS_GELU>(p, 1, stream);
      case EPI_BIAS_DROP_RESID: return launch<false, false, EPI_BIAS_DROP_RESID>(p, 1, stream);
      case EPI_ADD_F32: return launch<false, false, EPI_ADD_F32>(p, 1, stream);
    }
  } else if (form == CAREL_GEMM_NN) {
    switch (epi) {
      case EPI_BIAS_BF16: return launch<false, true, EPI_BIAS_BF16>(p, 1, stream);
      case EPI_DGELU_BF16: return launch<false, true, EPI_DGELU_BF16>(p, 1, stream);
      case EPI_ADD_F32: return launch<false, true, EPI_ADD_F32>(p, 1, stream);
    }
  } else if (form == CAREL_GEMM_TN) {
    if (epi == EPI_SLAB_F32) return launch<true, true, EPI_SLAB_F32>(p, splits, stream);
  }
  return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: unsupported form/epilogue combination (%d,%d)", form, epi);
}

extern "C" int carel_slab_reduce_f32(const void* slabs, void* out, long n, int splits, int accumulate,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!slabs || !out || n <= 0 || (n & 3) || splits <= 0)
    return set_error(CAREL_ERR_ARG, "carel_slab_reduce_f32: bad arguments (n must be a multiple of 4)");
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)slabs,
                     (float*)out, n, splits, accumulate);
  return check_launch("slab_reduce_kernel");
}
