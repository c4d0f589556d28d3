// EXPERIMENTS build only (-DCAREL_EXPERIMENTS, hook 342): built, bit-identical to the other kernels, measured, NOT adopted -- DESIGN.md 4.6.
// Small-M bf16 MFMA GEMM for the row-major-A forms (forward NT, data gradient NN) of the encoder linears at PACKED row counts
// (~1.7 k rows: ECPE batches with the padding skipped) -- the same nn.Linear calls under drl_classifier_ec_mmd_final_mul.py:202-206 / :841
// that gemm.hip / gemm_pp.hip serve at 8192 rows.
//
// Why a third kernel: at these sizes a GEMM is one short round of dependent K tiles per workgroup, and half of its time is
// fixed cost.  The 256 x 96 ping-pong tile keeps two K tiles (88 KB) in flight and walks 12 K tiles in ~7 us; the 128 x 128 kernel of
// gemm.hip keeps ONE in flight behind a vmcnt(0) + barrier per K step with one wave per SIMD.  Here:
//   * tile 128 x 128 x 64, 512 threads = 8 waves (4 x 2), wave tile 32 x 64 = 2 x 4 MFMA 16x16x32 accumulators: two waves per SIMD, so one
//     wave's fragment reads run under the other's MFMAs without any hand-written schedule;
//   * a FOUR-stage LDS-DMA ring (4 x 32 KB; one workgroup per CU -- the grids are <= 256 workgroups anyway): three K tiles in flight, counted
//     s_waitcnt vmcnt (every wave issues exactly 4 copies per K tile), ONE barrier per K tile;
//   * the K loop adds in the same order as the other two kernels (k ascending, 32 at a time): the same bits;
//   * epilogue through the LDS in two 64-column halves and the shared epi_store8 (every fused epilogue, split-K slabs included).
#ifdef CAREL_EXPERIMENTS
#include <type_traits>
#include "gemm_epilogue.h"
#include "reduce_device.h"

namespace carel {
namespace {

constexpr int SM_NST = 4, SM_STAGE = 32768, SM_LDS = SM_NST * SM_STAGE;

template <bool BT, int EPI>
__global__ __launch_bounds__(512) void gemm_sm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave & 3, wc = wave >> 2;
  int tm, tn;
  {  // each XCD (bid % 8) walks a contiguous chunk of the row-major tile order
    const int bid = blockIdx.x, nwg = p.tiles_m * p.tiles_n;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int tid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    tm = tid / p.tiles_n; tn = tid - tm * p.tiles_n;
  }
  const long m0 = (long)tm * 128, n0 = (long)tn * 128;
  const long kbase = (long)blockIdx.z * p.K;

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave's two 1-KiB pieces of each image (16 pieces per 16-KiB image; layouts of carel_common.h)
  uint32_t aoff[2], boff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = wave * 2 + i;
    {
      const int r = q * 8 + (lane >> 3), c = (lane & 7) ^ (r & 7);
      aoff[i] = (uint32_t)(((long)r * p.lda + c * 8) * 2);
    }
    if (BT) {
      const int r = q * 4 + (lane >> 4), c = (lane & 15) ^ swz_col(r);
      boff[i] = (uint32_t)(((long)r * p.ldb + c * 8) * 2);
    } else {
      const int r = q * 8 + (lane >> 3), c = (lane & 7) ^ (r & 7);
      boff[i] = (uint32_t)(((long)r * p.ldb + c * 8) * 2);
    }
  }
  const char* abase = (const char*)(p.A + m0 * p.lda + kbase);
  const char* bbase = (const char*)(BT ? p.B + kbase * p.ldb + n0 : p.B + n0 * p.ldb + kbase);
  const long bstep = BT ? 64 * p.ldb * 2 : 128;
  auto stage = [&](int kt) {
    char* ta = smem + (kt & (SM_NST - 1)) * SM_STAGE;
    char* tb = ta + 16384;
    const char* ag = abase + (long)kt * 128;
    const char* bg = bbase + (long)kt * bstep;
#pragma unroll
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((const void*)(ag + aoff[i]), (CAREL_LDS void*)(ta + (wave * 2 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((const void*)(bg + boff[i]), (CAREL_LDS void*)(tb + (wave * 2 + i) * 1024), 16, 0, 0);
    asm volatile("" ::: "memory");
  };

  const int nk = p.K >> 6;
  // Software pipeline over K tiles: the fragments of tile kt + 1 are requested from the LDS BEFORE the MFMAs of tile kt run (two fragment sets in
  // registers, the loop unrolled by two so that their indices are static): the LDS reads of one tile run under the matrix work of the previous one
  // inside every wave -- behind one barrier per K tile both waves of a SIMD would otherwise read together and then multiply together.
  bf16x8 fa[2][2][2], fb[2][2][4];          // [set][k32 step][block]
  auto read_frags = [&](auto SET, int kt) {
    constexpr int st = decltype(SET)::value;
    const char* ta = smem + (kt & (SM_NST - 1)) * SM_STAGE;
    const char* tb = ta + 16384;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[st][k2][i] = frag16_row(ta, wr * 32 + i * 16, k2 * 32);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) fb[st][k2][jj] = BT ? frag16_col(tb, wc * 64 + jj * 16, k2 * 32) : frag16_row(tb, wc * 64 + jj * 16, k2 * 32);
    }
  };
  auto mfmas = [&](auto SET) {
    constexpr int st = decltype(SET)::value;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = mfma16(fb[st][k2][jj], fa[st][k2][i], acc[i][jj]);      // swapped: each lane's 4 registers are 4 consecutive columns of one row
  };
  // one step: tile kt's fragments are in set SET; make tile kt + 1 ready in the other set, then multiply tile kt
  auto step = [&](auto SET, int kt) {
    constexpr int st = decltype(SET)::value;
    if (kt + 1 < nk) {
      // tile kt + 1 has landed once at most the copies of the ONE later tile in flight (kt + 2) are outstanding (vmcnt retires in order)
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                       // every wave's share of tile kt + 1 is in the LDS; every wave has its fragments of tile kt - 1 consumed
      if (kt + SM_NST - 1 < nk) stage(kt + SM_NST - 1);    // ... so that tile's stage may be overwritten (stage (kt - 1) % 4)
      read_frags(std::integral_constant<int, 1 - st>{}, kt + 1);
    }
    mfmas(SET);
  };
#pragma unroll
  for (int t = 0; t < SM_NST - 1; ++t)
    if (t < nk) stage(t);
  if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nk == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  read_frags(std::integral_constant<int, 0>{}, 0);
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    step(std::integral_constant<int, 0>{}, kt);
    step(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (kt < nk) step(std::integral_constant<int, 0>{}, kt);

  // ------------------------------------------------------------------ epilogue (gemm_kernel's, for 512 threads and 32-row wave tiles)
  float* ct = (float*)smem;                 // [128][CT_LD] fp32
  constexpr int CT_LD = 68;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
    if (wc == h) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *(f32x4*)(ct + (wr * 32 + i * 16 + (lane & 15)) * CT_LD + j * 16 + (lane >> 4) * 4) = acc[i][j];
    }
    __syncthreads();
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int r = it * 64 + (threadIdx.x >> 3), c8 = (threadIdx.x & 7) * 8;
      const float4 a = *(const float4*)(ct + r * CT_LD + c8), b = *(const float4*)(ct + r * CT_LD + c8 + 4);
      float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      epi_store8<EPI>(p, v, m0 + r, n0 + h * 64 + c8);
      if (epi_is_dgelu(EPI)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += v[e];
      }
    }
    if (epi_is_dgelu(EPI) && p.colsum_part) {       // block-uniform branch
      __syncthreads();
      float* sc = ct;                                     // [64][64] partial column sums
#pragma unroll
      for (int e = 0; e < 8; ++e) sc[(threadIdx.x >> 3) * 64 + (threadIdx.x & 7) * 8 + e] = cs[e];
      __syncthreads();
      if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) t += sc[r * 64 + threadIdx.x];
        p.colsum_part[(long)tm * p.N + n0 + h * 64 + threadIdx.x] = t;
      }
    }
  }
}

template <bool BT, int EPI>
int launch_sm(const GemmParams& p, int splits, hipStream_t s) {
  static bool attr_set = false;             // (a race sets it twice: harmless)
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_sm_kernel<BT, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS) != hipSuccess)
      return set_error(CAREL_ERR_HIP, "gemm_sm: hipFuncSetAttribute failed");
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_sm_kernel<BT, EPI>), dim3(p.tiles_m * p.tiles_n, 1, splits), dim3(512), SM_LDS, s, p);
  return check_launch("gemm_sm_kernel");
}

}  // namespace

// Shapes: M, N multiples of 128, K (per z slice) a multiple of 64; one round of at most 256 workgroups is what the kernel is for (the
// caller decides); p.tiles_m / p.tiles_n count 128 x 128 tiles.  epi = EPI_SLAB_F32 with splits > 1: z slices of p.K each into fp32 slabs.
int gemm_sm_launch(const GemmParams& p, bool bt, int epi, int splits, hipStream_t s) {
  if (p.M % 128 || p.N % 128 || p.K % 64 || p.K < 64 || splits < 1) return set_error(CAREL_ERR_SHAPE, "gemm_sm_launch: shape (M=%d N=%d K=%d)", p.M, p.N, p.K);
  if (!bt) {
    switch (epi) {
      case EPI_BIAS_BF16: return launch_sm<false, EPI_BIAS_BF16>(p, splits, s);
      case EPI_BIAS_GELU: return launch_sm<false, EPI_BIAS_GELU>(p, splits, s);
      case EPI_BIAS_GELU_DG: return launch_sm<false, EPI_BIAS_GELU_DG>(p, splits, s);
      case EPI_BIAS_DROP_RESID: return launch_sm<false, EPI_BIAS_DROP_RESID>(p, splits, s);
      case EPI_ADD_F32: return launch_sm<false, EPI_ADD_F32>(p, splits, s);
      case EPI_SLAB_F32: return launch_sm<false, EPI_SLAB_F32>(p, splits, s);
    }
  } else {
    switch (epi) {
      case EPI_BIAS_BF16: return launch_sm<true, EPI_BIAS_BF16>(p, splits, s);
      case EPI_DGELU_BF16: return launch_sm<true, EPI_DGELU_BF16>(p, splits, s);
      case EPI_MUL_BF16: return launch_sm<true, EPI_MUL_BF16>(p, splits, s);
      case EPI_ADD_F32: return launch_sm<true, EPI_ADD_F32>(p, splits, s);
      case EPI_SLAB_F32: return launch_sm<true, EPI_SLAB_F32>(p, splits, s);
    }
  }
  return set_error(CAREL_ERR_ARG, "gemm_sm_launch: unsupported form/epilogue (%d,%d)", (int)bt, epi);
}

}  // namespace carel
#endif   // CAREL_EXPERIMENTS
