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
#include "gemm_epilogue.h"
#include "reduce_device.h"

namespace carel {

// Epilogue of an internally split-K GEMM: v = sum_z slabs[z][row][col..col+7], then the normal fused epilogue.
template <int EPI>
__global__ __launch_bounds__(256) void slab_epilogue_kernel(GemmParams p, const float* __restrict__ slabs, int splits) {
  const long chunk = (long)blockIdx.x * 256 + threadIdx.x;           // one thread per 8 consecutive columns
  const int cpr = p.N >> 3;
  if (chunk >= (long)p.M * cpr) return;
  const long row = chunk / cpr, col = (chunk - row * cpr) * 8;
  const long off = row * p.N + col, plane = (long)p.M * p.N;
  float v[8];
  {
    const float4 a = *(const float4*)(slabs + off), b = *(const float4*)(slabs + off + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  for (int z = 1; z < splits; ++z) {
    const float4 a = *(const float4*)(slabs + z * plane + off), b = *(const float4*)(slabs + z * plane + off + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  epi_store8<EPI>(p, v, row, col);
}

// DBG (timing ablations only, results are wrong): 1 = no epilogue stores, 2 = no global->LDS loads after the
// first tile, 3 = no MFMA.  0 = the real kernel.
template <bool AT, bool BT, int EPI, int DBG = 0>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware tile map (blocks that share an XCD: bid % 8, round-robin dispatch).  Default (xcd_n = 1): each XCD gets a
  // contiguous chunk of the row-major tile order, walked N-fastest; the weight gradients use 1 x 8 patches (xcd_n = 8: a
  // compact (tiles_m / xm) x (tiles_n / xn) patch).  Option xcd_n = 0 (carel_gemm_set_variant(24)): each XCD owns
  // tiles_m/8 whole tile rows and walks them M-fastest, which keeps its A panels in L2 and streams B once.  Measured
  // (tools/exp_walk_pmc.py, tools/exp_walk_step.py): fabric-side fetch per launch falls -- QKV forward 97 -> 55 MB, FFN1
  // forward 170 -> 75 MB, FFN2 dgrad 180 -> 126 MB -- but the GEMMs get 3-4 % SLOWER (530 -> 512 TFLOP/s in the step,
  // 10.86 -> 11.16 ms serial, 9.34 -> 9.40 ms overlapped): the misses it removes were not on the critical path, and
  // the M-fastest order spreads each moment's output rows over 8x more distinct 128-row panels.  Hence not the default.
  const int bid = blockIdx.x;
  int tm, tn;
  {
    const int xn = p.xcd_n, xm = xn > 0 ? 8 / xn : 8;
    if (xn > 1 && p.tiles_m % xm == 0 && p.tiles_n % xn == 0) {
      const int xcd = bid & 7, local = bid >> 3;
      const int pm = p.tiles_m / xm, pn = p.tiles_n / xn;
      (void)pm;
      tm = (xcd / xn) * pm + local / pn;
      tn = (xcd % xn) * pn + local % pn;
    } else if (xn == 0 && (p.tiles_m & 7) == 0) {
      // each XCD owns tiles_m/8 whole tile rows and walks them M-FASTEST: the ~64 tiles in flight on an XCD then span
      // all of its rows x a few columns, so its A panels stay in L2 for the whole launch and B streams through once
      const int xcd = bid & 7, local = bid >> 3, rows = p.tiles_m >> 3;
      tm = xcd * rows + local % rows;
      tn = local / rows;
    } else {
      const int nwg = p.tiles_m * p.tiles_n;
      const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
      const int tid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
      tm = tid / p.tiles_n; tn = tid - tm * p.tiles_n;
    }
  }
  const long m0 = (long)tm * 128, n0 = (long)tn * 128;
  const long kbase = (long)blockIdx.z * p.K;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // TN form, first tile column only: also accumulate sum_k A[k][m] (ones-vector B operand) -> bias gradient
  const bool do_cs = AT && p.colsum_a != nullptr && tn == 0;
  f32x4 acc1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const s16x8 ones_bits = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_bits);

  // per-lane byte offsets of this wave's four DMA pieces of each operand (constant over K) and the wave-uniform bases
  uint32_t aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aoff[i] = AT ? col_lane_off(p.lda, i) : row_lane_off(p.lda, i);
    boff[i] = BT ? col_lane_off(p.ldb, i) : row_lane_off(p.ldb, i);
  }
  const char* abase = (const char*)(AT ? p.A + kbase * p.lda + m0 : p.A + m0 * p.lda + kbase);
  const char* bbase = (const char*)(BT ? p.B + kbase * p.ldb + n0 : p.B + n0 * p.ldb + kbase);
  const long astep = AT ? 64 * p.lda * 2 : 128, bstep = BT ? 64 * p.ldb * 2 : 128;     // bytes per K step
  auto stage = [&](int kt, int buf) {
    char* ta = smem + buf * 32768;
    char* tb = ta + 16384;
    const long kk = (DBG == 9) ? 0 : kt;                                 // DBG 9: every block re-reads one L2-hot tile
    stage_image_fast(((DBG == 9) ? (const char*)p.A : abase) + kk * astep, aoff, ta);
    stage_image_fast(((DBG == 9) ? (const char*)p.B : bbase) + kk * bstep, boff, tb);
  };

  const int nk = p.K >> 6;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk && DBG != 2) stage(kt + 1, (kt + 1) & 1);
    const char* ta = smem + (kt & 1) * 32768;
    const char* tb = ta + 16384;
    if (DBG == 7 || DBG == 8) {
      // ablation: LDS fragment reads removed (7: all, 8: the B operand's) after the first K step
      static_assert(true, "");
      bf16x8 fa[4], fb[4];
      if (kt == 0 || DBG == 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = frag16_row(ta, wr * 64 + i * 16, 0);
      }
      if (kt == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = frag16_row(tb, wc * 64 + i * 16, 0);
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        if (DBG == 8 && k2 == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[i] = frag16_row(ta, wr * 64 + i * 16, 32);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
      }
    } else if (DBG >= 5) {
      // variant under test: issue the fragment reads of BOTH k sub-steps before the first MFMA
      bf16x8 fa[2][4], fb[2][4];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[k2][i] = AT ? frag16_col(ta, wr * 64 + i * 16, k2 * 32) : frag16_row(ta, wr * 64 + i * 16, k2 * 32);
          fb[k2][i] = BT ? frag16_col(tb, wc * 64 + i * 16, k2 * 32) : frag16_row(tb, wc * 64 + i * 16, k2 * 32);
        }
      if (DBG == 6) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb[k2][j], fa[k2][i], acc[i][j]);
      if (DBG == 6) __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
    for (int ks = 0; ks < 64; ks += 32) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = AT ? frag16_col(ta, wr * 64 + i * 16, ks) : frag16_row(ta, wr * 64 + i * 16, ks);
        fb[i] = BT ? frag16_col(tb, wc * 64 + i * 16, ks) : frag16_row(tb, wc * 64 + i * 16, ks);
      }
      if (DBG == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (DBG == 3) { asm volatile("" :: "v"(fb[j]), "v"(fa[i])); }
          else acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);   // swapped: D[n][m]
        }
      if (DBG == 4) __builtin_amdgcn_s_setprio(0);
      if (AT && do_cs && wc == 0) {                           // block-uniform x wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) acc1[i] = mfma16(ones, fa[i], acc1[i]);
      }
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (DBG == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[i][j]));
    return;
  }

  if (AT && do_cs && wc == 0 && lane < 16) {                 // D[n][m]: every row n holds the same sum; lane = m
#pragma unroll
    for (int i = 0; i < 4; ++i) p.colsum_a[(long)blockIdx.z * p.M + m0 + wr * 64 + i * 16 + lane] = acc1[i][0];
  }
  // ------------------------------------------------------------------ epilogue
  // The accumulators go through LDS (the staging buffers are free now) so that global memory sees whole
  // 128/256-byte row segments: two halves of 64 columns; lane -> fragment rows on the way in,
  // 8 threads -> one 64-column row segment on the way out.
  float* ct = (float*)smem;                 // [128][CT_LD] fp32
  constexpr int CT_LD = 68;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
    if (wc == h) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *(f32x4*)(ct + (wr * 64 + i * 16 + (lane & 15)) * CT_LD + j * 16 + (lane >> 4) * 4) = acc[i][j];
    }
    __syncthreads();
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 32 + (threadIdx.x >> 3), c8 = (threadIdx.x & 7) * 8;
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
      float* sc = ct;                                     // [32][64] partial column sums
#pragma unroll
      for (int e = 0; e < 8; ++e) sc[(threadIdx.x >> 3) * 64 + (threadIdx.x & 7) * 8 + e] = cs[e];
      __syncthreads();
      if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += sc[r * 64 + threadIdx.x];
        p.colsum_part[(long)tm * p.N + n0 + h * 64 + threadIdx.x] = t;
      }
    }
  }
}

// =============================================================================================
// "big" tile: 256 x 192 x 64, 512 threads = 8 waves (2 x 4), each wave 128 x 48 = 8 x 3 MFMA tiles.
// The 128x128 kernel's main loop runs at the L2->LDS (LDS-DMA) bandwidth of its tile (64 FLOP per staged byte:
// tools/ablate_gemm.py); this tile stages (256+192) x 64 x 2 B per 2*256*192*64 FLOP = 110 FLOP/B.  One workgroup per
// CU (2 x 56..64 KiB LDS), two waves per SIMD.  Used for the N = 2304 / 3072 GEMMs and the weight gradients.
// =============================================================================================
constexpr int BIG_STAGE = 65536;           // A 32 KiB + B up to 32 KiB (two COL images)
constexpr int BIG_LDS = 2 * BIG_STAGE;

template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_kernel_big(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int tid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tm = tid / p.tiles_n, tn = tid - tm * p.tiles_n;
  const long m0 = (long)tm * 256, n0 = (long)tn * 192;
  const long kbase = (long)blockIdx.z * p.K;

  f32x4 acc[8][3];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_cs = AT && p.colsum_a != nullptr && tn == 0;
  f32x4 acc1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const s16x8 ones_bits = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_bits);

  auto stage = [&](int kt, int buf) {
    char* ta = smem + buf * BIG_STAGE;
    char* tb = ta + 32768;
    const long k0 = kbase + (long)kt * 64;
    if (AT) {                                          // [64 k][256 m] = two COL images
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = wave * 4 + i, img = q >> 4, qi = q & 15;
        const int r = qi * 4 + (lane >> 4);
        const int c = (lane & 15) ^ swz_col(r);
        __builtin_amdgcn_global_load_lds(p.A + (k0 + r) * p.lda + m0 + img * 128 + c * 8, (CAREL_LDS void*)(ta + img * 16384 + qi * 1024), 16, 0, 0);
      }
    } else {                                           // ROW image of 256 rows
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = wave * 4 + i;
        const int r = q * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);
        __builtin_amdgcn_global_load_lds(p.A + (m0 + r) * p.lda + k0 + c * 8, (CAREL_LDS void*)(ta + q * 1024), 16, 0, 0);
      }
    }
    if (BT) {                                          // [64 k][192 n]: COL image 0 full, image 1 first 64 columns
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = wave * 4 + i, img = q >> 4, qi = q & 15;
        const int r = qi * 4 + (lane >> 4);
        const int c = (lane & 15) ^ swz_col(r);
        if (img == 0 || c < 8)
          __builtin_amdgcn_global_load_lds(p.B + (k0 + r) * p.ldb + n0 + img * 128 + c * 8, (CAREL_LDS void*)(tb + img * 16384 + qi * 1024), 16, 0, 0);
      }
    } else {                                           // ROW image of 192 rows
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int q = wave * 3 + i;
        const int r = q * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);
        __builtin_amdgcn_global_load_lds(p.B + (n0 + r) * p.ldb + k0 + c * 8, (CAREL_LDS void*)(tb + q * 1024), 16, 0, 0);
      }
    }
  };

  const int nk = p.K >> 6;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* ta = smem + (kt & 1) * BIG_STAGE;
    const char* tb = ta + 32768;
#pragma unroll
    for (int ks = 0; ks < 64; ks += 32) {
      bf16x8 fa[8], fb[3];
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = AT ? frag16_col(ta + wr * 16384, i * 16, ks) : frag16_row(ta, wr * 128 + i * 16, ks);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int n = wc * 48 + j * 16;
        fb[j] = BT ? frag16_col(tb + (n >> 7) * 16384, n & 127, ks) : frag16_row(tb, n, ks);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
      if (AT && do_cs && wc == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc1[i] = mfma16(ones, fa[i], acc1[i]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (AT && do_cs && wc == 0 && lane < 16) {
#pragma unroll
    for (int i = 0; i < 8; ++i) p.colsum_a[(long)blockIdx.z * p.M + m0 + wr * 128 + i * 16 + lane] = acc1[i][0];
  }
  // epilogue through LDS in four 48-column slabs: thread -> (column chunk c = t % 6 fixed, rows t/6 + 85*it)
  float* ct = (float*)smem;
  constexpr int CT_LD = 52;
  const int t = threadIdx.x, cchunk = t % 6, rbase = t / 6;
#pragma unroll 1
  for (int h = 0; h < 4; ++h) {
    __syncthreads();
    if (wc == h) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          *(f32x4*)(ct + (wr * 128 + i * 16 + (lane & 15)) * CT_LD + j * 16 + (lane >> 4) * 4) = acc[i][j];
    }
    __syncthreads();
    float cs[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs[0][e] = 0.f; cs[1][e] = 0.f; }
    if (rbase < 85) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = rbase + 85 * it;
        if (r < 256) {
          const float4 a = *(const float4*)(ct + r * CT_LD + cchunk * 8), b = *(const float4*)(ct + r * CT_LD + cchunk * 8 + 4);
          float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
          epi_store8<EPI>(p, v, m0 + r, n0 + h * 48 + cchunk * 8);
          if (epi_is_dgelu(EPI)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[r >> 7][e] += v[e];
          }
        }
      }
    }
    if (epi_is_dgelu(EPI) && p.colsum_part) {       // per-128-row partial column sums (bias gradient), block-uniform
      __syncthreads();
      float* sc = ct;                                     // [2][85][48]
      if (rbase < 85) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[(0 * 85 + rbase) * 48 + cchunk * 8 + e] = cs[0][e]; sc[(1 * 85 + rbase) * 48 + cchunk * 8 + e] = cs[1][e]; }
      }
      __syncthreads();
      if (t < 96) {
        const int half = t / 48, c = t % 48;
        float tot = 0.f;
        for (int r = 0; r < 85; ++r) tot += sc[(half * 85 + r) * 48 + c];
        p.colsum_part[((long)tm * 2 + half) * p.N + n0 + h * 48 + c] = tot;
      }
    }
  }
}

template <bool AT, bool BT, int EPI>
static int launch_big(GemmParams p, int splits, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel_big<AT, BT, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "gemm_kernel_big: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr = true;
  }
  p.tiles_m = p.M / 256; p.tiles_n = p.N / 192;
  dim3 grid(p.tiles_m * p.tiles_n, 1, splits);
  hipLaunchKernelGGL((gemm_kernel_big<AT, BT, EPI>), grid, dim3(512), BIG_LDS, s, p);
  return check_launch("gemm_kernel_big");
}

CAREL_TUNABLE(int, g_gemm_variant, 0);   // 0 auto, 1 force the 128x128 kernel, 2 force the 256x192 kernel, 3 force the ping-pong kernel (NT / NN)
CAREL_TUNABLE(int, g_xcd_n, 1);          // XCD tile layout (see gemm_kernel): 1 = row-major chunks (default), 0 = row bands walked M-fastest, 2/4/8 = patches

// Small-M GEMMs (packed ECPE batches: ~1.8 k tokens) launch only 84-170 workgroups of 12-48 K steps each on 256 CUs.
// With a workspace they are run split-K into fp32 slabs + one fused-epilogue pass instead.
// GemmParams::split_tile_factor (carel::gemm_bf16_ex, internal): the caller runs this many equal GEMMs side by side (the
// forward's half-batch chains), so the split factor is chosen as for ONE GEMM over all of their rows -- the same K
// partition, hence the same bits, as the single-chain forward.
CAREL_TUNABLE(int, g_auto_split_min_k, 768);   // carel_gemm_set_variant(130 / 131): internal split-K for K >= 1536 only (round-1 behaviour) / K >= 768
static int auto_splits(const GemmParams& p, size_t ws_bytes) {
  if (g_gemm_variant != 0 || p.K < g_auto_split_min_k) return 1;
  const int factor = p.split_tile_factor & 0xff;
  const int tiles = p.tiles_m * p.tiles_n * (factor > 0 ? factor : 1);
  // K = 768: only where the caller says the row count does not depend on the batch (GEMM_EX_FIXED_ROWS: the [CLS]-only last layer,
  // M = 128 whatever the batch) -- the K partition, and with it every bit of a sample's result, must not depend on how a batch is
  // sharded over ranks (tests/test_gpu_dp2.py); at 84+ tiles (packed ECPE batches) the slab pass costs more than it saves (measured)
  const bool one_row_tile = (p.split_tile_factor & GEMM_EX_FIXED_ROWS) && p.M <= 128;
  if (p.K < 1536 && !one_row_tile) return 1;
  int s = 1;
  // a K slice keeps >= 6 K steps (384); the one-row-tile grids (6-24 workgroups, each K step a full L2 / HBM round trip of ~1.5 us
  // with nothing else resident on the CU) go down to 3
  while (tiles * s < 256 && p.K >= 768 && (p.K / (s * 2)) >= (one_row_tile ? 192 : 384) && p.K % (128 * s) == 0 &&
         (size_t)(s * 2) * p.M * p.N * 4 <= ws_bytes) s *= 2;
  return s;
}

CAREL_TUNABLE(int, g_pp_split, 1);       // carel_gemm_set_variant(140 / 141): internally split NT / NN GEMMs on the 128x128 kernel / on the ping-pong kernel where it fits
CAREL_TUNABLE(int, g_pp_min_tiles, 96);  // carel_gemm_set_variant(50 + k): the ping-pong kernel runs grids of at least 32 * k tiles (round 4, packed ECPE step, tools/ab_ecpe.sh: 4.23 ms at 96, 4.25 at 128-160, 4.28-4.30 at 192, 4.32 at 64, 4.65 at 32)
CAREL_TUNABLE(int, g_pp_min_tiles_k768, 32);   // carel_gemm_set_variant(40 + k): the same floor for K <= 768 (12 K tiles: a 56-tile grid of the ping-pong kernel is one short round; packed ECPE step 4.21 ms at 32 against 4.26 at 64-96, tools/ab_ecpe_k768.sh)
// Packed batches, attention-output forward (K = 768, 56 tiles of the ping-pong kernel, dropout + residual epilogue of 24 k elements per workgroup): split
// along K into slabs instead, so that the epilogue runs row-parallel inside the LayerNorm kernel behind it (ln_fwd_slabs_kernel).  Hook 330 + s.
CAREL_TUNABLE(int, g_resid_split, 0);     // NOT adopted: packed ECPE step 4.157 -> 4.128 ms with 2 slabs (tools/ab_ecpe_resid_split.sh), but a K = 768 partition that depends on the row count makes a sample's bits depend on how the batch is sharded over ranks (tests/test_gpu_dp2.py: global MMD within 1e-5)
static int resid_split(const GemmParams& p, int epi) {
  if (g_resid_split < 2 || epi != EPI_BIAS_DROP_RESID || p.K != 768 || !p.splitk_ws || p.M <= 128 || (p.M % 128) || (p.N % 128) || (p.N % 96)) return 1;
  const long t1 = (long)((p.M + 255) / 256) * (p.N / 96);
  if (t1 >= 96 || t1 * g_resid_split > 256 || (size_t)g_resid_split * p.M * p.N * 4 > p.splitk_ws_bytes) return 1;
  return g_resid_split;
}
CAREL_TUNABLE(int, g_big_auto, 0);       // set by carel_gemm_set_variant(30/31): 0 = never pick the big tile automatically
static bool big_auto(const GemmParams& p, int splits) {
  if (!g_big_auto) return false;
  const long tiles = (long)(p.M / 256) * (p.N / 192) * splits;
  return tiles >= 192 && (p.N >= 2304 || splits > 1);
}

// Small-M kernel (gemm_sm.hip): takes the row-major-A GEMMs of packed batches whose 128 x 128 tile grid is one round of at most 256 workgroups.
// Returns 0 (not this kernel) or the number of K slices (1 = single pass; > 1: fp32 slabs + slab epilogue, K >= 1536 only -- a K = 768 partition that
// depended on the row count would make a sample's bits depend on the shard size).  Tuning hook 340 + m (experiments build): 0 off, 2 on.
CAREL_TUNABLE(int, g_sm_mode, 0);
static int sm_plan(const GemmParams& p, int epi, bool at) {
#ifndef CAREL_EXPERIMENTS
  return 0;                                  // measured, not adopted: the kernel exists in the experiments build only
#endif
  if (at || g_sm_mode != 2 || g_gemm_variant != 0 || epi == EPI_SLAB_F32) return 0;
  if (p.M % 128 || p.N % 128 || p.K % 64 || p.M > 4096 || p.M <= 128 || p.ldc != p.N) return 0;
  const int factor = (p.split_tile_factor & 0xff) > 0 ? (p.split_tile_factor & 0xff) : 1;
  const long tiles = (long)(p.M / 128) * (p.N / 128) * factor;
  if (tiles > 256) return 0;
  int s = 1;
  if (p.splitk_ws && p.K >= 1536) {
    for (int c = 4; c >= 2; --c)
      if (tiles * c <= 256 && p.K % (64 * c) == 0 && p.K / c >= 512 && (size_t)c * p.M * p.N * 4 <= p.splitk_ws_bytes) { s = c; break; }
  }
  return s;
}

static thread_local int tl_split_plan = 1;      // result of the last GEMM_EX_PLAN_ONLY pass (gemm_bf16_split_plan)
template <bool AT, bool BT, int EPI>
static int launch(const GemmParams& p, int splits, hipStream_t s) {
  const bool plan_only = (p.split_tile_factor & GEMM_EX_PLAN_ONLY) != 0, defer = (p.split_tile_factor & GEMM_EX_DEFER_EPILOGUE) != 0;
  if (plan_only) {          // the decisions below without a launch: does this call take the internal split-K path, and with how many slabs?
    tl_split_plan = 1;
    if (const int sms = sm_plan(p, EPI, AT)) { tl_split_plan = sms; return CAREL_OK; }
    if (!AT && g_gemm_variant == 0) {
      const bool only_pp = !((p.M % 128 == 0 && p.N % 128 == 0) || (p.M % 256 == 0 && p.N % 192 == 0));
      if (resid_split(p, EPI) > 1) { tl_split_plan = resid_split(p, EPI); return CAREL_OK; }
      if (gemm_pp_pick(p, BT, EPI, only_pp ? 1 : -(p.K <= 768 ? g_pp_min_tiles_k768 : g_pp_min_tiles))) return CAREL_OK;
      const bool big_ok = (p.M % 256 == 0) && (p.N % 192 == 0), v1_ok = (p.M % 128 == 0) && (p.N % 128 == 0);
      if ((big_ok && !v1_ok) || !v1_ok) return CAREL_OK;
      if (EPI != EPI_SLAB_F32 && p.splitk_ws) tl_split_plan = auto_splits(p, p.splitk_ws_bytes);
    }
    return CAREL_OK;
  }
#ifdef CAREL_EXPERIMENTS
  if (const int sms = sm_plan(p, EPI, AT)) {
    if (sms == 1) {
      if (defer) return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: internal: epilogue deferred for a GEMM that does not split (plan / launch disagree)");
      return gemm_sm_launch(p, BT, EPI, 1, s);
    }
    GemmParams q = p;
    q.K = p.K / sms; q.outf = p.splitk_ws; q.ldc = p.N; q.colsum_part = nullptr;
    const int rc = gemm_sm_launch(q, BT, EPI_SLAB_F32, sms, s);
    if (rc || defer) return rc;
    const long chunks = (long)p.M * (p.N >> 3);
    hipLaunchKernelGGL((slab_epilogue_kernel<EPI>), dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, p, (const float*)p.splitk_ws, sms);
    return check_launch("gemm_sm_kernel split-K + slab_epilogue_kernel");
  }
#endif
  if (!AT && g_gemm_variant != 1 && g_gemm_variant != 2 && !(g_gemm_variant >= 11 && g_gemm_variant <= 19)) {
    // row-major-A forms: the 256 x 96n ping-pong kernel (gemm_pp.hip) when the grid fills the chip (variant 3: always)
#ifdef CAREL_EXPERIMENTS
    if (p.pair_flags && p.ldc == p.N && !p.colsum_part && g_gemm_variant == 0 && gemm_pp_pick_pair(p, BT, EPI)) return gemm_pp_launch_pair(p, BT, EPI, s);
#endif
    // (a shape only the ping-pong kernel takes -- M not a multiple of 128 -- runs on it whatever the tile-count threshold says:
    // gemm_shape_ok accepted it on that kernel's account)
#ifdef CAREL_EXPERIMENTS
    if (!BT && g_gemm_variant == 0 && p.ldc == p.N && gemm_tri_pick(p, EPI)) return gemm_tri_launch(p, EPI, s);
#endif
    const bool only_pp = !((p.M % 128 == 0 && p.N % 128 == 0) || (p.M % 256 == 0 && p.N % 192 == 0));
    const int npn = (g_gemm_variant == 0 && resid_split(p, EPI) > 1) ? 0 :
                    gemm_pp_pick(p, BT, EPI, (g_gemm_variant == 3 || g_gemm_variant >= 60 || only_pp) ? 1 : -(p.K <= 768 ? g_pp_min_tiles_k768 : g_pp_min_tiles));
#ifdef CAREL_GEMM_ABLATE
    if (npn && !BT && EPI == EPI_BIAS_BF16 && g_gemm_variant >= 61 && g_gemm_variant <= 69) return gemm_pp_launch_dbg(p, npn, g_gemm_variant - 60, s);
#endif
    if (npn) return gemm_pp_launch(p, BT, EPI, npn, s);
  }
  const bool big_ok = (p.M % 256 == 0) && (p.N % 192 == 0);
  const bool v1_ok = (p.M % 128 == 0) && (p.N % 128 == 0);
  // big tile when asked for (variant 2) or, automatically, where it wins: wide outputs / weight gradients with enough
  // tiles to fill the chip (heuristic from tools/bench_gemm.py)
  bool use_big = big_ok && (g_gemm_variant == 2 || (g_gemm_variant == 0 && (!v1_ok || big_auto(p, splits))));
  if (use_big) return launch_big<AT, BT, EPI>(p, splits, s);
  if (!v1_ok) return set_error(CAREL_ERR_SHAPE, "carel_gemm_bf16: shape fits neither tile (M=%d N=%d)", p.M, p.N);
  dim3 grid(p.tiles_m * p.tiles_n, 1, splits);
#ifdef CAREL_GEMM_ABLATE     // timing ablations (wrong results) are not part of the product library: build with -DCAREL_GEMM_ABLATE
  if (!AT && !BT && EPI == EPI_BIAS_BF16 && g_gemm_variant >= 11 && g_gemm_variant <= 19) {
    if (g_gemm_variant == 11) hipLaunchKernelGGL((gemm_kernel<false, false, EPI_BIAS_BF16, 1>), grid, dim3(256), 0, s, p);
    if (g_gemm_variant == 12) hipLaunchKernelGGL((gemm_kernel<false, false, EPI_BIAS_BF16, 2>), grid, dim3(256), 0, s, p);
    if (g_gemm_variant == 13) hipLaunchKernelGGL((gemm_kernel<false, false, EPI_BIAS_BF16, 3>), grid, dim3(256), 0, s, p);
    if (g_gemm_variant == 14) hipLaunchKernelGGL((gemm_kernel<false, false, EPI_BIAS_BF16, 4>), grid, dim3(256), 0, s, p);
    if (g_gemm_variant == 15) hipLaunchKernelGGL((gemm_kernel<false, false, EPI_BIAS_BF16, 5>), grid, dim3(256), 0, s, p);
    if (g_gemm_variant == 16) hipLaunchKernelGGL((gemm_kernel<false, false, EPI_BIAS_BF16, 6>), grid, dim3(256), 0, s, p);
    if (g_gemm_variant == 17) hipLaunchKernelGGL((gemm_kernel<false, false, EPI_BIAS_BF16, 7>), grid, dim3(256), 0, s, p);
    if (g_gemm_variant == 18) hipLaunchKernelGGL((gemm_kernel<false, false, EPI_BIAS_BF16, 8>), grid, dim3(256), 0, s, p);
    if (g_gemm_variant == 19) hipLaunchKernelGGL((gemm_kernel<false, false, EPI_BIAS_BF16, 9>), grid, dim3(256), 0, s, p);
    return check_launch("gemm_kernel<dbg>");
  }
#endif
  if (EPI != EPI_SLAB_F32 && p.splitk_ws) {
    const int rs = (!AT && g_gemm_variant == 0) ? resid_split(p, EPI) : 1;
    const int sp = rs > 1 ? rs : auto_splits(p, p.splitk_ws_bytes);
    if (sp > 1) {
      GemmParams q = p;
      q.K = p.K / sp; q.outf = p.splitk_ws; q.ldc = p.N; q.colsum_part = nullptr;
      // the slices on the ping-pong kernel where its tiles x slices fit one round (packed ECPE batches, ~1.8 k rows: 56 tiles x 4 slices of
      // 12 K tiles each run in ~13 us against ~20 us for 84 x 4 workgroups of the 128x128 kernel with its barrier + vmcnt(0) per K step);
      // same K partition, same order of additions inside a slice: the same bits (tests/test_gpu_gemm.py)
      int pp_npn = 0;
      if (!AT && g_pp_split && p.N % 96 == 0 && p.M > 128 && (p.K / sp) % 64 == 0 && p.K / sp >= 256) {
        const long t1 = (long)((p.M + 255) / 256) * (p.N / 96) * sp;
        if (t1 <= 256) pp_npn = 1;
        else if (p.N % 192 == 0 && t1 / 2 <= 256) pp_npn = 2;
      }
      const long chunks = (long)p.M * (p.N >> 3);
      if (pp_npn) {
        q.K = p.K;                       // the ping-pong kernel slices K by gridDim.z itself
        int rc = gemm_pp_launch_slab(q, BT, pp_npn, sp, s);
        if (rc) return rc;
        if (defer) return CAREL_OK;          // the caller's next kernel consumes the slabs (GEMM_EX_DEFER_EPILOGUE)
        hipLaunchKernelGGL((slab_epilogue_kernel<EPI>), dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, p, (const float*)p.splitk_ws, sp);
        return check_launch("gemm_pp_kernel split-K + slab_epilogue_kernel");
      }
      hipLaunchKernelGGL((gemm_kernel<AT, BT, EPI_SLAB_F32>), dim3(p.tiles_m * p.tiles_n, 1, sp), dim3(256), 0, s, q);
      if (defer) return check_launch("gemm_kernel split-K (epilogue deferred)");
      hipLaunchKernelGGL((slab_epilogue_kernel<EPI>), dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, p, (const float*)p.splitk_ws, sp);
      return check_launch("gemm_kernel split-K + slab_epilogue_kernel");
    }
  }
  if (defer) return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: internal: epilogue deferred for a GEMM that does not split (plan / launch disagree)");
  hipLaunchKernelGGL((gemm_kernel<AT, BT, EPI>), grid, dim3(256), 0, s, p);
  return check_launch("gemm_kernel");
}

// Sum `splits` fp32 slabs [splits][n] -> out[n] (+ optional accumulate into out)
__device__ __forceinline__ void slab_reduce_segment(const float* __restrict__ slabs, float* __restrict__ out, long n, int splits, int accumulate, int nblk) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)nblk * blockDim.x * 4;
  for (; i < n; i += stride) {
    float4 a = *(const float4*)(slabs + i);
    int z = 1;
    for (; z + 3 < splits; z += 4) {        // four slabs requested before the first add (same order of additions: same bits)
      float4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = *(const float4*)(slabs + (long)(z + u) * n + i);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
    }
    for (; z < splits; ++z) {
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
// One launch reduces up to three things (same order of additions as separate launches: same bits):
//   the weight-gradient slabs [splits][n] -> out;
//   slabs2 != null: a small [splits][n2] array -> out2 -- the bias-gradient partials the weight-gradient GEMM leaves behind its slabs
//     (they used to cost a launch of their own: 24 launches of ~5 us per step);
//   pc.partials != null: the blocks past `nblk` sum the per-block partials of a LayerNorm backward into dgamma, dbeta and the bias gradient
//     (partial_reduce_seg_kernel of ln.hip: the same device function) -- 72 latency-bound blocks that used to be a launch of their own
//     in front of every FFN2 / out-projection weight gradient (24 per step).
struct PartCols { const float* partials; SegOuts outs; int seg, n, nparts; };
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, long n, int splits, int accumulate,
                                                          const float* __restrict__ slabs2, float* __restrict__ out2, long n2, int nblk, PartCols pc) {
  __shared__ float lds[256];
  if ((int)blockIdx.x < nblk) {
    slab_reduce_segment(slabs, out, n, splits, accumulate, nblk);
    if (slabs2) slab_reduce_segment(slabs2, out2, n2, splits, 0, nblk);
    return;
  }
  int c;
  const float t = partial_colsum16(pc.partials, pc.n, pc.nparts, lds, c, (int)blockIdx.x - nblk);
  if (threadIdx.x < PR_COLS && c < pc.n) {
    float* o = pc.outs.p[c / pc.seg];
    if (o) o[c % pc.seg] = t;
  }
}


}  // namespace carel

using namespace carel;

// ---- optional per-launch HIP-event timing of the GEMM kernel (bench.py's roofline leg) -------------
#include <algorithm>
#include <vector>
namespace {
struct GemmProf {
  bool on = false;
  std::vector<hipEvent_t> ev;      // start/stop pairs
  std::vector<double> flops;
  size_t used = 0;
  std::vector<hipEvent_t> cal;     // 16 start/stop pairs around an empty kernel: the bracket's own cost
  std::vector<hipEvent_t> cal2;    // 16 start/stop pairs with nothing in between: what the two event records alone cost
  double last_empty_us = 0.0, last_pair_us = 0.0;
  bool calibrated = false;
} g_prof;
__global__ void prof_empty_kernel() {}
struct ProfScope {
  hipStream_t s; bool active;
  ProfScope(hipStream_t st, double fl) : s(st), active(false) {
    if (!g_prof.on || fl < 0.0 || g_prof.used + 2 > g_prof.ev.size()) return;
    if (!g_prof.calibrated) {        // once, on the stream being profiled: what an event pair around a kernel costs by itself
      g_prof.calibrated = true;
      for (size_t i = 0; i + 1 < g_prof.cal.size(); i += 2) {
        (void)hipEventRecord(g_prof.cal[i], s);
        hipLaunchKernelGGL(prof_empty_kernel, dim3(1), dim3(64), 0, s);
        (void)hipEventRecord(g_prof.cal[i + 1], s);
      }
      for (size_t i = 0; i + 1 < g_prof.cal2.size(); i += 2) {
        (void)hipEventRecord(g_prof.cal2[i], s);
        (void)hipEventRecord(g_prof.cal2[i + 1], s);
      }
    }
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
  for (hipEvent_t e : g_prof.cal) (void)hipEventDestroy(e);
  for (hipEvent_t e : g_prof.cal2) (void)hipEventDestroy(e);
  g_prof.ev.clear(); g_prof.cal.clear(); g_prof.cal2.clear(); g_prof.flops.clear(); g_prof.used = 0; g_prof.on = false; g_prof.calibrated = false;
  if (!enable) return CAREL_OK;
  if (max_launches < 1) return set_error(CAREL_ERR_ARG, "carel_profile_gemm: max_launches must be positive");
  g_prof.ev.resize((size_t)max_launches * 2);
  g_prof.cal.resize(32); g_prof.cal2.resize(32);
  for (auto& e : g_prof.cal2)
    if (hipEventCreate(&e) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_profile_gemm: hipEventCreate failed");
  for (auto& e : g_prof.ev)
    if (hipEventCreate(&e) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_profile_gemm: hipEventCreate failed");
  for (auto& e : g_prof.cal)
    if (hipEventCreate(&e) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_profile_gemm: hipEventCreate failed");
  g_prof.on = true;
  return CAREL_OK;
}

// Host-synchronising read-out: sums the recorded launches (ms, algorithmic flops 2*M*N*K, count).  The time an event
// pair around an EMPTY kernel takes on the same stream (median of 16, measured at the first profiled launch) is
// subtracted from every bracket, so the sum is kernel time as `rocprofv3 --kernel-trace` reports it.
extern "C" int carel_profile_gemm_read(double* total_ms, double* total_flops, int64_t* launches) {
  if (!total_ms || !total_flops || !launches) return set_error(CAREL_ERR_ARG, "carel_profile_gemm_read: null output");
  double ms = 0.0, fl = 0.0;
  double overhead = 0.0;
  if (g_prof.calibrated) {
    std::vector<float> c;
    for (size_t i = 0; i + 1 < g_prof.cal.size(); i += 2) {
      float t = 0.f;
      if (hipEventSynchronize(g_prof.cal[i + 1]) == hipSuccess && hipEventElapsedTime(&t, g_prof.cal[i], g_prof.cal[i + 1]) == hipSuccess) c.push_back(t);
    }
    if (!c.empty()) { std::sort(c.begin(), c.end()); overhead = c[c.size() / 2]; }
    g_prof.last_empty_us = overhead * 1e3;
    std::vector<float> c2;
    for (size_t i = 0; i + 1 < g_prof.cal2.size(); i += 2) {
      float t = 0.f;
      if (hipEventSynchronize(g_prof.cal2[i + 1]) == hipSuccess && hipEventElapsedTime(&t, g_prof.cal2[i], g_prof.cal2[i + 1]) == hipSuccess) c2.push_back(t);
    }
    if (!c2.empty()) { std::sort(c2.begin(), c2.end()); g_prof.last_pair_us = c2[c2.size() / 2] * 1e3; }
  }
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_profile_gemm_read: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_profile_gemm_read: elapsed failed");
    ms += (t > overhead ? t - overhead : 0.0); fl += g_prof.flops[i / 2];
  }
  *total_ms = ms; *total_flops = fl; *launches = (int64_t)(g_prof.used / 2);
  g_prof.used = 0; g_prof.flops.clear();
  return CAREL_OK;
}

// The two calibration medians of the last carel_profile_gemm_read (us): an event pair around an EMPTY kernel (what _read subtracts from
// every bracket) and an event pair with nothing in between.  bracket - empty = kernel time over an empty kernel's; bracket - pair = the
// whole interval the stream spends on the launch, i.e. what `rocprofv3 --kernel-trace` reports as the duration plus the dispatch gap.
extern "C" int carel_profile_gemm_overheads(double* empty_kernel_bracket_us, double* event_pair_us) {
  if (!empty_kernel_bracket_us || !event_pair_us) return set_error(CAREL_ERR_ARG, "carel_profile_gemm_overheads: null output");
  *empty_kernel_bracket_us = g_prof.last_empty_us; *event_pair_us = g_prof.last_pair_us;
  return CAREL_OK;
}

static int gemm_shape_ok(int M, int N, int K, int splits, int form) {
  if (M <= 0 || N <= 0 || K <= 0 || splits <= 0) return 0;
  if (form == CAREL_GEMM_TN) {                                          // ping-pong kernel: unequal K slices are fine
    if (M % 256 == 0 && N % 96 == 0 && K % 64 == 0 && (K >> 6) / splits >= 4) return 1;
  }
  const bool pp = form != CAREL_GEMM_TN && N % 96 == 0 && K >= 256;     // ping-pong kernel: any M (edge rows masked)
  if (!((M % 128 == 0 && N % 128 == 0) || (M % 256 == 0 && N % 192 == 0) || pp)) return 0;
  if (K % (64 * splits)) return 0;
  return 1;
}

// tuning hook (carel_gemm_set_variant(190 + m)): 0 = never fuse LayerNorm into the 768-wide linears (default: measured), 1 = out-projection and FFN2,
// 2 = out-projection only
#ifdef CAREL_EXPERIMENTS
CAREL_TUNABLE(int, g_rowln_mode, 0);
CAREL_TUNABLE(long, g_rowln_min_rows, 6144);        // 192 workgroups of 32 rows: below that the chip is not filled
namespace carel {
int gemm_rowln_wanted(long rows) { return g_rowln_mode != 0 && rows >= g_rowln_min_rows; }
int gemm_rowln_wanted_k(int K) { return g_rowln_mode == 1 || K <= 768; }
}
extern "C" int carel_gemm_rowln(const carel_gemm_rowln_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a) return set_error(CAREL_ERR_ARG, "carel_gemm_rowln: null args");
  if (a->M < 1 || a->K < 128 || (a->K & 127)) return set_error(CAREL_ERR_SHAPE, "carel_gemm_rowln: M >= 1 and K a multiple of 128 (M=%d K=%d)", a->M, a->K);
  if (!a->A || !a->W || !a->resid_f32 || !a->gamma || !a->beta) return set_error(CAREL_ERR_ARG, "carel_gemm_rowln: null operand");
  if ((((uintptr_t)a->A | (uintptr_t)a->W | (uintptr_t)a->resid_f32 | (uintptr_t)a->gamma | (uintptr_t)a->beta | (uintptr_t)a->bias | (uintptr_t)a->h_f32 |
        (uintptr_t)a->x_f32 | (uintptr_t)a->x_bf16) & 15) || (a->lda & 7) || a->lda < a->K || (!a->w_packed && ((a->ldb & 7) || a->ldb < a->K)))
    return set_error(CAREL_ERR_ARG, "carel_gemm_rowln: operands must be 16-byte aligned, leading dimensions multiples of 8 and >= K");
  RowLnParams p;
  p.A = (const bf16_t*)a->A; p.lda = a->lda; p.W = (const bf16_t*)a->W; p.ldb = a->ldb; p.bias = (const float*)a->bias; p.resid = (const float*)a->resid_f32;
  p.gamma = (const float*)a->gamma; p.beta = (const float*)a->beta; p.eps = a->eps;
  p.h_out = (float*)a->h_f32; p.x_f32 = (float*)a->x_f32; p.x_bf16 = (bf16_t*)a->x_bf16; p.stats = (float*)a->stats; p.M = a->M; p.K = a->K;
  p.drop = make_dropout(a->drop_seed, a->drop_site, a->drop_p, a->drop_idx_offset); p.drop_row_map = (const int*)a->drop_row_map;
  ProfScope prof_scope(stream, 2.0 * (double)a->M * 768.0 * (double)a->K);
  return gemm_rowln_launch(p, a->w_packed != 0, stream);
}
extern "C" int carel_gemm_rowln_pack(const void* W, int64_t ldb, int32_t K, void* out, void* stream) {
  if (!W || !out || K < 128 || (K & 127) || ldb < K || (ldb & 7) || (((uintptr_t)W | (uintptr_t)out) & 15))
    return set_error(CAREL_ERR_ARG, "carel_gemm_rowln_pack: bad arguments (K multiple of 128, ldb >= K multiple of 8, 16-byte aligned pointers)");
  return gemm_rowln_pack(W, (long)ldb, K, out, (hipStream_t)stream);
}

// EXPERIMENTS build only: the product library has no tuning hooks (every switch below is a compile-time constant there)
extern "C" int carel_gemm_set_variant(int32_t v) {
  if (v >= 20 && v <= 23) { g_xcd_n = 1 << (v - 20); return CAREL_OK; }    // 20: 8x1, 21: 4x2, 22: 2x4, 23: 1x8
  if (v == 24) { g_xcd_n = 0; return CAREL_OK; }                            // 24: XCD row bands walked M-fastest
  if (v == 30 || v == 31) { g_big_auto = v - 30; return CAREL_OK; }
  if (v >= 50 && v <= 59) { g_pp_min_tiles = (v - 50) * 32; return CAREL_OK; }
  if (v >= 340 && v <= 342) { g_sm_mode = v - 340; return CAREL_OK; }              // small-M kernel (gemm_sm.hip) off / - / takes packed-batch GEMMs of <= 256 tiles
  if (v >= 330 && v <= 333) { g_resid_split = v - 330; return CAREL_OK; }
  if (v >= 40 && v <= 49) { g_pp_min_tiles_k768 = (v - 40) * 32; return CAREL_OK; }
  if (v >= 70 && v <= 73) { gemm_pp_force_npn(v - 70); return CAREL_OK; }
  if (v >= 100 && v <= 116) { gemm_pp_wgrad_force(v - 100); return CAREL_OK; }         // ping-pong weight gradient: split-K factor forced (0 = heuristic)
  if (v == 130 || v == 131) { g_auto_split_min_k = v == 130 ? 1536 : 768; return CAREL_OK; }
  if (v == 140 || v == 141) { g_pp_split = v - 140; return CAREL_OK; }
  if (v == 160 || v == 161) { gemm_pp_gelu_lut(v - 160); return CAREL_OK; }
  if (v >= 190 && v <= 192) { g_rowln_mode = v - 190; return CAREL_OK; }
  if (v >= 193 && v <= 195) { gemm_rowln_dbg(v - 193); return CAREL_OK; }                 // (ablation builds: 194 no MFMA, 195 no weight loads)
  if (v == 270 || v == 271) { encoder_ln_slab_fusion_enable(v - 270); return CAREL_OK; }   // split-K slab epilogues as their own launch / fused into the following LayerNorm (default)
  if (v >= 250 && v <= 252) { gemm_pp_group_mode(v - 250); return CAREL_OK; }             // grouped weight gradients: 256 x 96 tiles + split remainder (default) / 256 x 192 whole / 256 x 96 whole
  if (v == 240 || v == 241) { encoder_wgrad_group_enable(v - 240); return CAREL_OK; }   // one GEMM + reduction per weight gradient / one grouped launch per layer (default)
  if (v == 230 || v == 231) { encoder_ln_resid_enable(v - 230); return CAREL_OK; }     // encoder forward: LayerNorm f32 outputs stored and re-read / recomputed by the next residual epilogue (default)
  if (v >= 220 && v <= 225) { gemm_tri_enable(v - 220); return CAREL_OK; }             // three-group kernel for the N = 768 forward GEMMs off (default) / on
  if (v >= 280 && v <= 292) { adam_grid_cap(32L << (v - 280)); return CAREL_OK; }         // Adam launch: at most 32 << k workgroups (292: 131072 = one float4 per thread, default)
  if (v == 210 || v == 211) { tail_overlap_enable(v - 210); return CAREL_OK; }           // VAE tail: loss kernel on the side stream beside the decoder passes off / on (default)
  if (v == 200 || v == 201) { gemm_pp_pair_enable(v - 200); return CAREL_OK; }        // pair split-K of the N = 768, K >= 1536 GEMMs off (default: measured slower) / on
  if (v == 170 || v == 171) { gemm_pp_epi_prefetch(v - 170); return CAREL_OK; }       // ping-pong kernel: epilogue inputs requested before the main loop off / on
  if (v == 120 || v == 121) { gemm_pp_xcd_rect(v - 120); return CAREL_OK; }             // ping-pong kernel, NT / NN: XCD tile map chunks / rectangles
  if (v == 90 || v == 91) { gemm_pp_wide_variant(v - 90); return CAREL_OK; }           // wide-phase schedule of the ping-pong kernel (npn 2) off / on
  g_gemm_variant = v;
  return CAREL_OK;
}
#endif   // CAREL_EXPERIMENTS

extern "C" int carel_gemm_bf16(const carel_gemm_args* a, void* stream_) { return carel::gemm_bf16_ex(a, 1, stream_); }

int carel::gemm_bf16_split_plan(const carel_gemm_args* a, int split_tile_factor) {
  tl_split_plan = 1;
  if (carel::gemm_bf16_ex(a, (split_tile_factor & ~GEMM_EX_DEFER_EPILOGUE) | GEMM_EX_PLAN_ONLY, nullptr)) return 1;
  return tl_split_plan;
}

int carel::gemm_bf16_ex(const carel_gemm_args* a, int split_tile_factor, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a) return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: null args");
  const int splits = a->splits > 0 ? a->splits : 1;
  if (!gemm_shape_ok(a->M, a->N, a->K, splits, a->form))
    return set_error(CAREL_ERR_SHAPE, "carel_gemm_bf16: (M,N) must be multiples of (128,128) or (256,192) and K of 64*splits (M=%d N=%d K=%d splits=%d)",
                     a->M, a->N, a->K, splits);
  if (!a->A || !a->B) return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: null operand");
  if (((uintptr_t)a->A | (uintptr_t)a->B) & 15 || (a->lda & 7) || (a->ldb & 7) || (a->ldc & 3))
    return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: operands must be 16-byte aligned, ld multiples of 8");
  GemmParams p = {};
  p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.lda = a->lda; p.ldb = a->ldb;
  p.M = a->M; p.N = a->N; p.K = a->K / splits;
  p.out0 = (bf16_t*)a->out_bf16; p.out1 = (bf16_t*)a->out2_bf16; p.outf = (float*)a->out_f32; p.ldc = a->ldc;
  p.bias = (const float*)a->bias; p.resid = (const float*)a->resid_f32; p.aux = (const bf16_t*)a->aux_bf16;
  p.resid_stats = (const float*)a->resid_ln_stats; p.resid_gamma = (const float*)a->resid_ln_gamma; p.resid_beta = (const float*)a->resid_ln_beta;
  if (p.resid_stats || p.resid_gamma || p.resid_beta) {
    if (a->epilogue != EPI_BIAS_DROP_RESID || !p.resid_stats || !p.resid_gamma || !p.resid_beta)
      return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: resid_ln_stats / _gamma / _beta go together and with CAREL_EPI_BIAS_DROP_RESID only");
    if ((((uintptr_t)p.resid_stats) & 7) || (((uintptr_t)p.resid_gamma | (uintptr_t)p.resid_beta) & 15))
      return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: resid_ln_* pointers must be 8 / 16-byte aligned");
  }
  p.drop = make_dropout(a->drop_seed, a->drop_site, a->drop_p, a->drop_idx_offset);
  p.tiles_m = a->M / 128; p.tiles_n = a->N / 128;
  p.split_tile_factor = split_tile_factor;
  p.xcd_n = (a->form == CAREL_GEMM_TN && g_xcd_n <= 1) ? 8 : g_xcd_n;   // wgrad: 1x8 patches measured best (tools/bench_gemm.py)
  p.splitk_ws = (float*)a->splitk_ws; p.splitk_ws_bytes = a->splitk_ws ? (size_t)a->splitk_ws_bytes : 0;
  p.pair_flags = nullptr; p.pair_seq = 0;
  if (a->splitk_ws && a->splitk_ws_zeroed && (size_t)a->splitk_ws_bytes > (size_t)(64 << 20) && ((uintptr_t)a->splitk_ws & 15) == 0 && (a->splitk_ws_bytes & 15) == 0) {
    // the last 4 KiB of a zero-initialised workspace hold the pair split-K flags; every other user of the workspace sees it shorter
    p.splitk_ws_bytes -= PP_PAIR_FLAG_BYTES;
    p.pair_flags = (unsigned*)((char*)a->splitk_ws + p.splitk_ws_bytes);
  }
  if (a->ldc != a->N) p.splitk_ws = nullptr;                    // the slab epilogue assumes a dense C
  if (a->colsum_part) p.splitk_ws = nullptr;                    // fused column sums live in the single-pass epilogue
  p.colsum_part = (float*)a->colsum_part; p.drop_row_map = (const int*)a->drop_row_map; p.colsum_a = (float*)a->colsum_a;
  if (p.colsum_a && a->form != CAREL_GEMM_TN) return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: colsum_a is a TN-form (wgrad) option");
  const int form = a->form, epi = a->epilogue;
  if (splits != 1 && epi != EPI_SLAB_F32) return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: split-K only with the slab epilogue");
#define NEED(ptr, what) if (!(ptr)) return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: epilogue needs " what)
  switch (epi) {
    case EPI_BIAS_BF16: NEED(p.out0, "out_bf16"); break;
    case EPI_BIAS_GELU: NEED(p.out1, "out2_bf16"); break;                  // out_bf16 (the pre-activation) is optional
    case EPI_BIAS_GELU_DG: NEED(p.out0, "out_bf16"); NEED(p.out1, "out2_bf16"); break;
    case EPI_BIAS_DROP_RESID: NEED(p.outf, "out_f32"); NEED(p.resid, "resid_f32"); break;
    case EPI_DGELU_BF16: case EPI_MUL_BF16: NEED(p.out0, "out_bf16"); NEED(p.aux, "aux_bf16"); break;
    case EPI_ADD_F32: NEED(p.outf, "out_f32"); break;
    case EPI_SLAB_F32: NEED(p.outf, "out_f32"); break;
    default: return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: unknown epilogue %d", epi);
  }
#undef NEED
  ProfScope prof_scope(stream, (split_tile_factor & GEMM_EX_PLAN_ONLY) ? -1.0 : 2.0 * (double)a->M * (double)a->N * (double)a->K);
  if (form == CAREL_GEMM_NT) {
    switch (epi) {
      case EPI_BIAS_BF16: return launch<false, false, EPI_BIAS_BF16>(p, 1, stream);
      case EPI_BIAS_GELU: return launch<false, false, EPI_BIAS_GELU>(p, 1, stream);
      case EPI_BIAS_GELU_DG: return launch<false, false, EPI_BIAS_GELU_DG>(p, 1, stream);
      case EPI_BIAS_DROP_RESID: return launch<false, false, EPI_BIAS_DROP_RESID>(p, 1, stream);
      case EPI_ADD_F32: return launch<false, false, EPI_ADD_F32>(p, 1, stream);
    }
  } else if (form == CAREL_GEMM_NN) {
    switch (epi) {
      case EPI_BIAS_BF16: return launch<false, true, EPI_BIAS_BF16>(p, 1, stream);
      case EPI_DGELU_BF16: return launch<false, true, EPI_DGELU_BF16>(p, 1, stream);
      case EPI_MUL_BF16: return launch<false, true, EPI_MUL_BF16>(p, 1, stream);
      case EPI_ADD_F32: return launch<false, true, EPI_ADD_F32>(p, 1, stream);
    }
  } else if (form == CAREL_GEMM_TN) {
    if (epi == EPI_SLAB_F32) {
      if (g_gemm_variant != 1 && g_gemm_variant != 2) {      // ping-pong kernel: K tiles dealt to the slices as evenly as possible
        GemmParams q = p;
        q.K = a->K;
        const int npn = gemm_pp_pick_tn(q, splits);
        const long wgs = npn ? (long)(q.M / 256) * (q.N / (96 * npn)) * splits : 0;
        // (the same floor as gemm_pp_wgrad_splits: whatever carel_gemm_wgrad_splits proposes for this kernel must be taken by it)
#ifdef CAREL_GEMM_ABLATE
        if (npn == 2 && g_gemm_variant >= 61 && g_gemm_variant <= 68) return gemm_pp_launch_tn_dbg(q, npn, splits, g_gemm_variant - 60, stream);
#endif
        if (npn && (g_gemm_variant == 3 || wgs >= 64)) return gemm_pp_launch_tn(q, npn, splits, stream);
      }
      if (a->K % (64 * splits) || !((a->M % 128 == 0 && a->N % 128 == 0) || (a->M % 256 == 0 && a->N % 192 == 0)))
        return set_error(CAREL_ERR_SHAPE, "carel_gemm_bf16: this (M,N,K,splits) fits neither weight-gradient kernel (M=%d N=%d K=%d splits=%d)", a->M, a->N, a->K, splits);
      return launch<true, true, EPI_SLAB_F32>(p, splits, stream);
    }
  }
  return set_error(CAREL_ERR_ARG, "carel_gemm_bf16: unsupported form/epilogue combination (%d,%d)", form, epi);
}

// Split-K factor for the weight gradient dW[M,N] = dY^T X over T tokens (the slab buffer must hold that many [M][N] planes,
// plus [splits][M] floats when the bias gradient rides along).  Ping-pong kernel: enough slices to fill the 256 CUs.
// 128x128 kernel (variant 1, or shapes the ping-pong kernel does not take): measured at T = 8192 in the whole step with
// the weight gradients on the side stream -- 36 tiles (768x768) -> 8, 108 tiles (2304x768) -> 4, 144 tiles (FFN) -> 4.
static int wgrad_splits_128(int M, int N, long T) {
  const int tiles = (M / 128) * (N / 128);
  const int want = tiles < 64 ? 8 : 4;
  int s = 1;
  while (s < want && T % (64L * s * 2) == 0 && T / (s * 2) >= 256) s *= 2;
  return s;
}
extern "C" int32_t carel_gemm_wgrad_splits(int32_t M, int32_t N, int64_t T) {
  if (g_gemm_variant != 1 && g_gemm_variant != 2) {
    const int s = gemm_pp_wgrad_splits(M, N, (long)T);
    if (s > 0) return s;
  }
  return wgrad_splits_128(M, N, (long)T);
}
namespace carel {
// the largest factor carel_gemm_wgrad_splits can return for this shape under ANY tuning-hook setting (buffer sizing)
int gemm_wgrad_splits_max(int M, int N, long T) {
  const int a = gemm_pp_wgrad_splits(M, N, T), b = wgrad_splits_128(M, N, T);
  // hook 100 + s may force up to min(16, K tiles / 4) slices on the ping-pong kernel AFTER the slab buffer was sized
  const long nk4 = (T >> 6) / 4;
  const int c = (M % 256 == 0 && N % 96 == 0 && T % 64 == 0 && T >= 512) ? (int)(nk4 < 16 ? nk4 : 16) : 0;
  const int ab = a > b ? a : b;
  return ab > c ? ab : c;
}
}  // namespace carel

// ---- grouped weight gradients (gemm_pp.hip) --------------------------------------------------------------------------------------------
static int wgrad_group_probs(const carel_wgrad_group_args* a, WgradGroupProb* pb) {
  if (!a || a->n_prob < 1 || a->n_prob > 4) return 0;
  for (int i = 0; i < a->n_prob; ++i) pb[i] = WgradGroupProb{a->prob[i].dY, a->prob[i].X, a->prob[i].dW, a->prob[i].db, a->prob[i].M, a->prob[i].N};
  return a->n_prob;
}
extern "C" int64_t carel_gemm_wgrad_group_ws_bytes(const carel_wgrad_group_args* a) {
  WgradGroupProb pb[4];
  const int n = wgrad_group_probs(a, pb);
  if (!n || !gemm_pp_wgrad_group_ok(pb, n, (long)a->T)) return -1;
  return (int64_t)gemm_pp_wgrad_group_ws_bytes(pb, n, (long)a->T);
}
extern "C" int carel_gemm_wgrad_group(const carel_wgrad_group_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  WgradGroupProb pb[4];
  const int n = wgrad_group_probs(a, pb);
  if (!n) return set_error(CAREL_ERR_ARG, "carel_gemm_wgrad_group: null args or n_prob outside 1..4");
  if (a->n_ln < 0 || a->n_ln > 2) return set_error(CAREL_ERR_ARG, "carel_gemm_wgrad_group: n_ln outside 0..2");
  WgradGroupLn ln[2];
  for (int i = 0; i < a->n_ln; ++i) {
    if (!a->ln[i].partials || a->ln[i].rows < 1) return set_error(CAREL_ERR_ARG, "carel_gemm_wgrad_group: null LayerNorm partials");
    ln[i] = WgradGroupLn{a->ln[i].partials, carel_layernorm_bwd_blocks(a->ln[i].rows), a->ln[i].dgamma, a->ln[i].dbeta, a->ln[i].dbias};
  }
  double fl = 0.0;
  for (int i = 0; i < n; ++i) fl += 2.0 * (double)pb[i].M * (double)pb[i].N * (double)a->T;
  ProfScope prof_scope(stream, fl);         // (the bracket covers the reduction launch too)
  return gemm_pp_wgrad_group(pb, n, (long)a->T, a->workspace, a->workspace_bytes > 0 ? (size_t)a->workspace_bytes : 0, ln, a->n_ln, stream);
}

extern "C" int carel_slab_reduce_f32(const void* slabs, void* out, long n, int splits, int accumulate,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!slabs || !out || n <= 0 || (n & 3) || splits <= 0)
    return set_error(CAREL_ERR_ARG, "carel_slab_reduce_f32: bad arguments (n must be a multiple of 4)");
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)slabs,
                     (float*)out, n, splits, accumulate, (const float*)nullptr, (float*)nullptr, 0L, (int)blocks, PartCols{});
  return check_launch("slab_reduce_kernel");
}
namespace carel {
// out[n] = sum_z slabs[z][n], optionally out2[n2] = sum_z slabs2[z][n2] (slabs2 may be null) and optionally the column sums of LayerNorm-backward
// partials [nparts][3 * 768] -> (dgamma, dbeta, dbias) (partials may be null), all in ONE launch
int slab_reduce_multi(const void* slabs, void* out, int64_t n, const void* slabs2, void* out2, int64_t n2, int splits, const void* partials,
                      int nparts, void* dgamma, void* dbeta, void* dbias, hipStream_t stream) {
  if (!slabs || !out || n <= 0 || (n & 3) || splits < 1 || (slabs2 && (!out2 || n2 <= 0 || (n2 & 3)))) return set_error(CAREL_ERR_ARG, "slab_reduce_multi: bad arguments");
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  PartCols pc{};
  int extra = 0;
  if (partials) {
    pc.partials = (const float*)partials; pc.outs.p[0] = (float*)dgamma; pc.outs.p[1] = (float*)dbeta; pc.outs.p[2] = (float*)dbias; pc.outs.p[3] = nullptr;
    pc.seg = 768; pc.n = 3 * 768; pc.nparts = nparts;
    extra = (3 * 768 + PR_COLS - 1) / PR_COLS;
  }
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(blocks + extra)), dim3(256), 0, stream, (const float*)slabs, (float*)out, (long)n, splits, 0,
                     (const float*)slabs2, (float*)out2, (long)n2, (int)blocks, pc);
  return check_launch("slab_reduce_kernel");
}
}
