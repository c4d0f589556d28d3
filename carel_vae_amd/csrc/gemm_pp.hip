// Ping-pong bf16 MFMA GEMM for the row-major-A forms of the encoder linears (forward NT, data gradient NN) of CAREL-VAE:
// replaces nn.Linear.forward / the dX half of its backward inside HF BertSelfAttention / BertSelfOutput /
// BertIntermediate / BertOutput (drl_classifier_ec_mmd_final_mul.py:202-206, :841).
//
// Why a second kernel: the 128x128 kernel (gemm.hip) stages 64 FLOP per LDS-DMA byte and its main loop runs at the
// L2->LDS rate of a CU (~65 GB/s, MI355X_MICROARCH.md "Indexed rows"), with one barrier + vmcnt(0) per K step.  Here:
//   * macro tile 256 x (96 * NPN), NPN = 1..3, one 512-thread workgroup per CU.  96 divides every encoder width
//     (768 / 2304 / 3072), so at M = 8192 the tile counts are exactly 256 (N = 768 with NPN = 1, N = 2304 with NPN = 3)
//     or 512 (N = 3072 with NPN = 2): no partial last round on the 256 CUs.  110-135 FLOP per staged byte at NPN 2-3.
//   * 8 waves = 4 (M) x 2 (N); wave tile 64 x 48*NPN = 2 M-halves x 2 x (3*NPN) MFMA 16x16x32 accumulators.
//   * the two wave columns are the two PING-PONG GROUPS (waves w and w+4 share a SIMD): group 1 runs one barrier behind
//     group 0, so while one group issues its 12 MFMAs of a phase the other issues the next phase's fragment reads and
//     DMA -- the matrix pipe of every SIMD always has one wave feeding it.
//   * LDS-DMA stays in flight ACROSS the barriers: raw s_barrier, counted s_waitcnt vmcnt(N) from a static schedule
//     (gemm_pp_sched.inc, generated and hazard-checked by tools/gemm_sched.py; rules in its docstring).  Never vmcnt(0) in
//     the steady state.
//   * a K tile (64 deep) = 2*NPN phases (M half) x (B part of 96 columns) in serpentine order, so each phase re-reads only
//     the operand that changed: 4 (A) or 6 (B) ds_read_b128 per 12 MFMAs.
//   * epilogue straight from the accumulators: v_permlane16_swap pairs two 16-column fragments so that every lane owns 8
//     consecutive columns of one row -> 16-byte bf16 stores / 2 x 16-byte f32 accesses, no LDS round trip, no barrier.
//
// DMA units (all exactly 2 global_load_lds_dwordx4 per wave, which is what makes the vmcnt immediates static):
//   A_h   rows {wr*64 + h*32 .. +32 | wr = 0..3} x 64 k   16 KiB of the stage's 256-row ROW image
//   B_j   NT: 96 weight rows x 64 k = 12 KiB ROW image (second instruction half-populated)
//         NN: 64 k-rows x 96 columns in a 256-B-pitch COL image = 16 KiB (12 of 16 chunks per row populated)
//   image row r of B_j <-> tile column (r / 48) * 48*NPN + j*48 + r % 48, i.e. each wave's columns are contiguous.
#include <utility>
#include "gemm_epilogue.h"

namespace carel {

#include "gemm_pp_sched.inc"

namespace {

template <int V> struct IC { static constexpr int value = V; };
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int PP_A_BYTES = 32768;
#ifndef CAREL_PP_MPRIO
#define CAREL_PP_MPRIO 1           // s_setprio level of the matrix segment (experiment: CAREL_EXTRA_FLAGS=-DCAREL_PP_MPRIO=0)
#endif
template <int NPN, bool BT> struct PPGeom {
  static constexpr int BPART = BT ? 16384 : 12288;
  static constexpr int STAGE = PP_A_BYTES + NPN * BPART;
  static constexpr int LDS = PPSched<NPN>::STAGES * STAGE;
};

// DBG (timing ablations, results wrong, only instantiated in a -DCAREL_GEMM_ABLATE build): 1 no DMA after the prologue,
// 2 no MFMA, 3 no fragment reads after the first tile, 4 no epilogue, 5 the half-populated second B instruction dropped
// WIDE: the schedule PPSchedW<NPN> -- one phase per B part with BOTH M halves (24 MFMAs per matrix segment instead of 12, half the
// barriers per K tile; the A fragments of both halves are read in phase 0 and kept: +16 registers), every wave drains its LDS reads
// before the barrier that ends its load segment (so a slot may be restaged ONE phase after its last read; tools/gemm_sched.py war = 1)
template <int NPN, bool AT, bool BT, int EPI, int DBG = 0, bool WIDE = false>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmParams p) {
  static_assert(!AT || BT, "the A^T form (weight gradient) has both operands K-strided");
  using S = std::conditional_t<WIDE, PPSchedW<NPN>, PPSched<NPN>>;
  using G = PPGeom<NPN, BT>;
  static_assert(S::STAGES == PPSched<NPN>::STAGES, "PPGeom sizes the LDS from the fine schedule's stage count");
  constexpr int NP = S::NP, ST = S::STAGES;
  constexpr int BN = 96 * NPN, WN = 48 * NPN, NF = 3 * NPN;
  constexpr int BPART = G::BPART, STAGE = G::STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave & 3, wc = wave >> 2;                     // wc = ping-pong group

  // XCD-aware tile map: blocks with equal bid % 8 share an XCD (round-robin dispatch; speed only); each XCD walks a
  // contiguous chunk of the row-major tile order (bijective for any tile count)
  // One tile per workgroup.  (A persistent loop over tiles was built and measured: the next tile's first counted vmcnt
  // wait then absorbs the previous tile's store acknowledgements, and those arrive at the HBM write rate -- an XCD's 32 CUs
  // write more per round than its L2 holds -- so nothing overlapped, and the loop-carried state cost 40-60 VGPRs.)
  int tm, tn;
  {
    const int nwg = p.tiles_m * p.tiles_n, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int tid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    tm = tid / p.tiles_n; tn = tid - tm * p.tiles_n;
  }
  const long m0 = (long)tm * 256, n0 = (long)tn * BN;

  // ---- per-lane DMA source offsets (bytes, constant over K) and wave-uniform LDS destinations ------------------------
  uint32_t aoff[2][2];                                         // [M half][instruction]
  int adst[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!AT) {                                               // ROW image of 256 rows; half h = rows wr*64 + h*32 .. +32
        const int piece = wr * 8 + h * 4 + wc * 2 + k;         // 8-row piece
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (lane >> 3);                // source chunk = physical chunk ^ (row & 7)
        long rg = m0 + row; if (rg > (long)p.M - 1) rg = (long)p.M - 1;     // rows past M re-read the last row (never stored)
        aoff[h][k] = (uint32_t)(((rg - m0) * p.lda + c * 8) * 2);
        adst[h][k] = piece * 1024;
      } else {                                                 // two COL images [64 k][128 m]; half h = image h = m h*128 .. +128
        const int q = wave * 2 + k;                            // 4 k-rows per piece
        const int r = q * 4 + (lane >> 4);
        const int c = (lane & 15) ^ swz_col(r);
        aoff[h][k] = (uint32_t)(((long)r * p.lda + h * 128 + c * 8) * 2);
        adst[h][k] = h * 16384 + q * 1024;
      }
    }
  uint32_t boff[2];
  int bdst[2];
  bool bact[2];
  if (!BT) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int base_row = k == 0 ? wave * 8 : (8 + (wave >> 1)) * 8 + (wave & 1) * 4;
      const int r = base_row + (lane >> 3);                    // k = 1: lanes 0..31 only (4 rows)
      const int c = (lane & 7) ^ (r & 7);
      boff[k] = (uint32_t)((((long)(r / 48) * WN + r % 48) * p.ldb + c * 8) * 2);
      bdst[k] = base_row * 128;
      bact[k] = k == 0 || lane < 32;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = wave * 2 + k;                              // 4 k-rows per piece
      const int r = q * 4 + (lane >> 4);
      const int c = (lane & 15) ^ swz_col(r);                  // logical 8-column chunk 0..11 (12..15 unused)
      boff[k] = (uint32_t)(((long)r * p.ldb + (c / 6) * WN + (c % 6) * 8) * 2);
      bdst[k] = q * 1024;
      bact[k] = c < 12;
    }
  }
  // K range of this z slice: the K tiles are dealt to gridDim.z slices as evenly as possible (slices may differ by one)
  const int nk_all = p.K >> 6;
  const int kt0 = (int)(((long)blockIdx.z * nk_all) / gridDim.z), kt1 = (int)(((long)(blockIdx.z + 1) * nk_all) / gridDim.z);
  const int nk = kt1 - kt0;
  const long a_step = AT ? 64 * p.lda * 2 : 128, b_step = BT ? 64 * p.ldb * 2 : 128;
  const char* a_ptr = (const char*)(AT ? p.A + m0 : p.A + m0 * p.lda) + kt0 * a_step;      // first K tile of the slice
  const char* b_ptr = (const char*)(BT ? p.B + n0 : p.B + n0 * p.ldb) + kt0 * b_step;
  const long b_part_step = BT ? 48 * 2 : 48 * p.ldb * 2;       // part j -> j + 1

  // issue unit `u` (0/1 = A halves, 2+j = B parts) of K tile (t + d) into LDS stage `stg`; ap / bp = pointers of tile t
  auto issue = [&](auto U, const char* ap, const char* bp, int d, int stg) {
    constexpr int u = decltype(U)::value;
    char* sb = smem + stg * STAGE;
    if constexpr (u < 2) {
      const char* g = ap + (long)d * a_step;
      __builtin_amdgcn_global_load_lds((const void*)(g + aoff[u][0]), (CAREL_LDS void*)(sb + adst[u][0]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void*)(g + aoff[u][1]), (CAREL_LDS void*)(sb + adst[u][1]), 16, 0, 0);
    } else {
      constexpr int j = u - 2;
      const char* g = bp + (long)d * b_step + j * b_part_step;
      char* pb = sb + PP_A_BYTES + j * BPART;
      // A compiler fence after every lane-masked instruction: LLVM (ROCm 7.2) otherwise merges the masked copy of one piece with the
      // unmasked copy of a neighbouring one into a single instruction whose LDS base is a per-lane select, and takes that base with
      // v_readfirstlane -- the upper half-wave then lands on the lower half's destination (seen with three units in one phase).
      if (bact[0]) __builtin_amdgcn_global_load_lds((const void*)(g + boff[0]), (CAREL_LDS void*)(pb + bdst[0]), 16, 0, 0);
      asm volatile("" ::: "memory");
      if (bact[1] && DBG != 5) __builtin_amdgcn_global_load_lds((const void*)(g + boff[1]), (CAREL_LDS void*)(pb + bdst[1]), 16, 0, 0);
      if (DBG == 5) __builtin_amdgcn_global_load_lds((const void*)(g + boff[0]), (CAREL_LDS void*)(pb + bdst[0]), 16, 0, 0);   // keeps the vmcnt arithmetic
    }
    asm volatile("" ::: "memory");     // and no instruction crosses a unit (vmcnt retires in issue order)
  };

  f32x4 acc[2][2][NF];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[WIDE ? 2 : 1][2][2], fb[3][2];                     // [M half (wide only)][16-row block][k32 step], [16-column block][k32 step]
  // A^T form, first tile column, group 0: sum_k A[k][m] through a ones-vector MFMA -> the bias gradient (colsum_a)
  const bool do_cs = AT && p.colsum_a != nullptr && tn == 0 && wc == 0;     // wave-uniform
  f32x4 acc1[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc1[h][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const s16x8 ones_bits = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_bits);

  // ---- prologue: the units the steady-state schedule would have issued before phase 0 ------------------------------
  static_for<S::NPRO>([&](auto I) {
    constexpr int i = decltype(I)::value;
    issue(IC<S::pro_unit[i]>{}, a_ptr, b_ptr, S::pro_tile[i], S::pro_tile[i] % ST);
  });
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::PRO_WAIT) : "memory");
  __builtin_amdgcn_s_barrier();
  if (wc == 1) __builtin_amdgcn_s_barrier();                   // group 1 runs one barrier behind group 0

  int sidx = 0;                                                // LDS stage of the current K tile
  bool first_tile = true;                                      // (ablation builds only)
  // one K tile; R = 0: steady state, R = r > 0: r tiles remain including this one (tail vmcnt tables, no issue past K)
  auto tile = [&](auto RR) {
    constexpr int R = decltype(RR)::value;
    const char* st = smem + sidx * STAGE;
    static_for<NP>([&](auto PP) {
      constexpr int P = decltype(PP)::value;
      constexpr int h = S::phase_h[P], j = S::phase_j[P];
      // ---------------- load segment L(P): fragments of this phase, this phase's DMA units, counted wait -------------
      const bool do_reads = (DBG != 3 && DBG != 6 && DBG != 7 && DBG != 8) || first_tile;     // 6: DMA + barriers only, 7: barriers only, 8: MFMA + barriers only
      if constexpr (WIDE) {
        if constexpr (P == 0) {
          if (do_reads)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int ks = 0; ks < 2; ++ks)
                fa[hh][i][ks] = AT ? frag16_col(st + hh * 16384, wr * 32 + i * 16, ks * 32) : frag16_row(st, wr * 64 + hh * 32 + i * 16, ks * 32);
        }
      } else if constexpr (P == 0 || S::phase_h[P] != S::phase_h[P == 0 ? 0 : P - 1]) {
        if (do_reads)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            fa[0][i][ks] = AT ? frag16_col(st + h * 16384, wr * 32 + i * 16, ks * 32) : frag16_row(st, wr * 64 + h * 32 + i * 16, ks * 32);
      }
      if constexpr (P == 0 || S::phase_j[P] != S::phase_j[P == 0 ? 0 : P - 1]) {
        const char* pb = st + PP_A_BYTES + j * BPART;
        if (do_reads)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            fb[jj][ks] = BT ? frag16_col(pb, wc * 48 + jj * 16, ks * 32) : frag16_row(pb, wc * 48 + jj * 16, ks * 32);
      }
      static_for<S::n_issue[P]>([&](auto E) {
        constexpr int e = decltype(E)::value;
        constexpr int u = S::issue_unit[P][e], d = S::issue_delta[P][e];
        if constexpr ((R == 0 || d < R) && DBG != 1 && DBG != 7 && DBG != 8) {
          int stg = sidx + d;
          if (stg >= ST) stg -= ST;
          issue(IC<u>{}, a_ptr, b_ptr, d, stg);
        }
      });
      if constexpr (S::wait[R][P] >= 0 && DBG != 1 && DBG != 7 && DBG != 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::wait[R][P]) : "memory");
      if constexpr (WIDE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads complete before the barrier: war = 1
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- matrix segment M(P) --------------------------------------------------------------------------
      __builtin_amdgcn_s_setprio(CAREL_PP_MPRIO);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hh = (WIDE ? 0 : h); hh < (WIDE ? 2 : h + 1); ++hh)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
              if (DBG == 2 || DBG == 6 || DBG == 7) asm volatile("" ::"v"(fb[jj][ks]), "v"(fa[WIDE ? hh : 0][i][ks]));
              else acc[hh][i][j * 3 + jj] = mfma16(fb[jj][ks], fa[WIDE ? hh : 0][i][ks], acc[hh][i][j * 3 + jj]);   // swapped: D[n][m]
            }
      if constexpr (AT && j == 0) {
        if (do_cs) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hh = (WIDE ? 0 : h); hh < (WIDE ? 2 : h + 1); ++hh)
#pragma unroll
              for (int i = 0; i < 2; ++i) acc1[hh][i] = mfma16(ones, fa[WIDE ? hh : 0][i][ks], acc1[hh][i]);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    });
    a_ptr += a_step; b_ptr += b_step;
    sidx = sidx + 1 == ST ? 0 : sidx + 1;
    first_tile = false;
  };
  for (int t = 0; t < nk - S::NTAIL; ++t) tile(IC<0>{});
  static_for<S::NTAIL>([&](auto I) { tile(IC<S::NTAIL - decltype(I)::value>{}); });
  if (wc == 0) __builtin_amdgcn_s_barrier();                   // both groups have now passed the same number of barriers

  if (AT && do_cs && lane < 16) {                              // D[n][m]: every row n holds the same sum; lane = m
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) p.colsum_a[(long)blockIdx.z * p.M + m0 + h * 128 + wr * 32 + i * 16 + lane] = acc1[h][i][0];
  }
  if (DBG == 4) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) asm volatile("" ::"v"(acc[h][i][j]));
    return;
  }
  // ---- epilogue: accumulators -> fused epilogue, 8 consecutive columns per lane ---------------------------------------
  // Row blocks b = (h, i) of 16 rows; the inputs (residual / pre-GELU rows) of block b + 1 are requested before block b is
  // stored, the bias once per column group up front: no load ever queues behind a store of its own wave (see epi_in8).
  const int rho = lane >> 4;
  constexpr int NQ = NF / 2;
  float cs[NQ > 0 ? NQ : 1][8];
#pragma unroll
  for (int q = 0; q < (NQ > 0 ? NQ : 1); ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[q][e] = 0.f;
  float bias8[NQ > 0 ? NQ : 1][8];
#pragma unroll
  for (int q = 0; q < NQ; ++q) epi_bias8<EPI>(p, n0 + wc * WN + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8, bias8[q]);
  auto row_of = [&](int b) { return m0 + (AT ? (b >> 1) * 128 + wr * 32 : wr * 64 + (b >> 1) * 32) + (b & 1) * 16 + (lane & 15); };
  // (npn 3 with a residual / aux input has no registers for two blocks of inputs: load and use block by block there;
  // the encoder never runs that combination -- N = 2304 is the bias-only QKV projection)
  constexpr bool PIPE = NPN < 3 || EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU || EPI == EPI_SLAB_F32;
  EpiIn8 in[PIPE ? 2 : 1][NQ > 0 ? NQ : 1];
  auto load_block = [&](int b, EpiIn8* dst) {
    const long row = row_of(b);
    if (row < (long)p.M) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) epi_in8<EPI>(p, row, n0 + wc * WN + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8, dst[q]);
    }
  };
  if (PIPE) load_block(0, in[0]);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (PIPE) { if (b + 1 < 4) load_block(b + 1, in[(b + 1) & 1]); }
    else load_block(b, in[0]);
    const int h = b >> 1, i = b & 1;
    const long row = row_of(b);
    const bool ok = row < (long)p.M;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      // fragments 2q, 2q+1: after the swaps, 16-lane row rho holds fragment 2q + (rho & 1), columns (rho >> 1) * 8 .. +8
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[h][i][2 * q][e]), __float_as_uint(acc[h][i][2 * q + 1][e]), false, false);
        v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
      }
      const long col = n0 + wc * WN + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8;
      if (ok) {
        epi_out8<EPI>(p, v, bias8[q], in[PIPE ? (b & 1) : 0][q], row, col);
        if (EPI == EPI_DGELU_BF16) {
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[q][e] += v[e];
        }
      }
    }
  }
  if constexpr (NF & 1) {      // the odd last fragment: 4 columns per lane, after all paired stores
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const long row = row_of(b);
      if (row < (long)p.M) epi_store<EPI>(p, acc[b >> 1][b & 1][NF - 1], row, n0 + wc * WN + (NF - 1) * 16 + rho * 4);
    }
  }
  if (EPI == EPI_DGELU_BF16 && p.colsum_part) {                // block-uniform; dispatcher guarantees NF even here
    // per-128-row column sums of the stored values (the FFN1 bias gradient): 16 lanes -> 1, then wave rows 2r, 2r+1
    float* sc = (float*)smem;                                  // [4 wave rows][BN]; every LDS read / DMA has retired
#pragma unroll
    for (int q = 0; q < NF / 2; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = cs[q][e];
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
        if ((lane & 15) == 0) sc[wr * BN + wc * WN + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8 + e] = t;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // raw barriers: a __syncthreads here would also drain the stores
    __builtin_amdgcn_s_barrier();
    for (int x = threadIdx.x; x < 2 * BN; x += 512) {
      const int half = x / BN, c = x - half * BN;
      const long prow = (long)tm * 2 + half;
      if (prow * 128 < (long)p.M) p.colsum_part[prow * p.N + n0 + c] = sc[(2 * half) * BN + c] + sc[(2 * half + 1) * BN + c];
    }
  }
}

// =====================================================================================================================
// Loader-wave variant (round 2, NT / NN forms, NPN 1-2): the same tile, phases and schedule, but the LDS-DMA is issued -- and its
// counted vmcnt waits are executed -- by TWO EXTRA WAVES (8, 9; each stands in for four compute waves' pieces, 8 instructions
// per unit, so every immediate of the schedule is x4), and the workgroup is PERSISTENT over its tiles on one continuous
// stream of K tiles.  Why: vmcnt retires in issue order, so a wave that both stores an epilogue and waits for DMA cannot let
// its stores drain behind the next tile's main loop (measured with the 8-wave persistent loop: nothing overlapped).  Here the
// eight compute waves never wait on vmcnt in the main loop: they fire a tile's epilogue stores and go straight on with the
// next tile, whose first K tiles the loader waves already have in flight; the store traffic (the HBM-bound part of a GEMM
// launch) leaves L2 under the next main loop.  The loaders keep group 0's barrier cadence, so the RAW / WAR rules and the
// tables of tools/gemm_sched.py hold unchanged.  M must be a multiple of 256 (no edge rows); grid = min(tiles, 256), a
// multiple of 8 when there is more than one round (a workgroup's tiles then stay in one XCD chunk).
// =====================================================================================================================
// NLW loader waves (2 or 4; 4 = one per SIMD, each standing in for two compute waves: a wave's LDS-DMA instruction costs it about 60
// cycles of issue, so two loaders alone cannot keep up with a 45-KiB K tile); WIDE as in gemm_pp_kernel.
template <int NPN, bool BT, int EPI, int NLW, bool WIDE>
__global__ __launch_bounds__(512 + 64 * NLW) void gemm_ppl_kernel(GemmParams p) {
  using S = std::conditional_t<WIDE, PPSchedW<NPN>, PPSched<NPN>>;
  using G = PPGeom<NPN, BT>;
  static_assert(S::STAGES == PPSched<NPN>::STAGES, "PPGeom sizes the LDS from the fine schedule's stage count");
  constexpr int EMU = 8 / NLW;                                            // compute waves' pieces per loader wave
  constexpr int NP = S::NP, ST = S::STAGES;
  constexpr int BN = 96 * NPN, WN = 48 * NPN, NF = 3 * NPN;
  constexpr int BPART = G::BPART, STAGE = G::STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // 0..7 compute, 8..9 loaders
  const int nk = p.K >> 6;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nmine = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int V = nmine * nk;                                               // K tiles of this workgroup's whole stream
  auto tile_rc = [&](int j, int& tm, int& tn) {                           // j-th tile of this workgroup (XCD-chunked order)
    const int bid = (int)blockIdx.x + j * (int)gridDim.x;
    const int xcd = bid & 7, qq = ntiles >> 3, rr = ntiles & 7;
    const int tid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    tm = tid / p.tiles_n; tn = tid - tm * p.tiles_n;
  };
  const long a_step = 128, b_step = BT ? 64 * p.ldb * 2 : 128;
  const long b_part_step = BT ? 48 * 2 : 48 * p.ldb * 2;

  if (wave >= 8) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - 8;
    uint32_t aoff[EMU][2][2], boff[EMU][2];
    int adst[EMU][2][2], bdst[EMU][2];
    bool bact[EMU][2];
#pragma unroll
    for (int i = 0; i < EMU; ++i) {
      const int w = lw * EMU + i, wr = w & 3, wc = w >> 2;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int piece = wr * 8 + h * 4 + wc * 2 + k;
          const int row = piece * 8 + (lane >> 3);
          const int c = (lane & 7) ^ (lane >> 3);
          aoff[i][h][k] = (uint32_t)(((long)row * p.lda + c * 8) * 2);
          adst[i][h][k] = piece * 1024;
        }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (!BT) {
          const int base_row = k == 0 ? w * 8 : (8 + (w >> 1)) * 8 + (w & 1) * 4;
          const int r = base_row + (lane >> 3);
          const int c = (lane & 7) ^ (r & 7);
          boff[i][k] = (uint32_t)((((long)(r / 48) * WN + r % 48) * p.ldb + c * 8) * 2);
          bdst[i][k] = base_row * 128;
          bact[i][k] = k == 0 || lane < 32;
        } else {
          const int q = w * 2 + k;
          const int r = q * 4 + (lane >> 4);
          const int c = (lane & 15) ^ swz_col(r);
          boff[i][k] = (uint32_t)(((long)r * p.ldb + (c / 6) * WN + (c % 6) * 8) * 2);
          bdst[i][k] = q * 1024;
          bact[i][k] = c < 12;
        }
      }
    }
    // operand pointers of the virtual K tiles v, v+1, v+2 (slot d = tile v + d); refreshed as v advances
    const char* va[3];
    const char* vb[3];
    auto locate = [&](int vv, const char*& a, const char*& b) {
      if (vv >= V) { a = (const char*)p.A; b = (const char*)p.B; return; }      // never issued (tail tables)
      const int j = vv / nk, kt = vv - j * nk;
      int tm, tn;
      tile_rc(j, tm, tn);
      a = (const char*)(p.A + (long)tm * 256 * p.lda) + kt * a_step;
      b = (const char*)(BT ? p.B + (long)tn * BN : p.B + (long)tn * BN * p.ldb) + kt * b_step;
    };
    locate(0, va[0], vb[0]); locate(1, va[1], vb[1]); locate(2, va[2], vb[2]);
    auto issue = [&](auto U, int d, int stg) {
      constexpr int u = decltype(U)::value;
      char* sb = smem + stg * STAGE;
#pragma unroll
      for (int i = 0; i < EMU; ++i) {
        if constexpr (u < 2) {
          __builtin_amdgcn_global_load_lds((const void*)(va[d] + aoff[i][u][0]), (CAREL_LDS void*)(sb + adst[i][u][0]), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const void*)(va[d] + aoff[i][u][1]), (CAREL_LDS void*)(sb + adst[i][u][1]), 16, 0, 0);
        } else {
          constexpr int j = u - 2;
          const char* g = vb[d] + j * b_part_step;
          char* pb = sb + PP_A_BYTES + j * BPART;
          if (bact[i][0]) __builtin_amdgcn_global_load_lds((const void*)(g + boff[i][0]), (CAREL_LDS void*)(pb + bdst[i][0]), 16, 0, 0);
          // Compiler fence after EVERY masked instruction.  Without it LLVM (ROCm 7.2) merges the masked and unmasked copies of
          // neighbouring pieces into one instruction whose LDS base is a per-lane select, and then takes that base with
          // v_readfirstlane: the upper half-wave lands on the lower half's destination (seen as stale B rows, round 2).
          asm volatile("" ::: "memory");
          if (bact[i][1]) __builtin_amdgcn_global_load_lds((const void*)(g + boff[i][1]), (CAREL_LDS void*)(pb + bdst[i][1]), 16, 0, 0);
          asm volatile("" ::: "memory");
        }
      }
      // vmcnt retires in ISSUE order and the schedule's immediates count whole units: no instruction may move across a unit
      asm volatile("" ::: "memory");
    };
    static_for<S::NPRO>([&](auto I) {
      constexpr int i = decltype(I)::value;
      issue(IC<S::pro_unit[i]>{}, S::pro_tile[i], S::pro_tile[i] % ST);
    });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EMU * S::PRO_WAIT) : "memory");
    __builtin_amdgcn_s_barrier();
    int sidx = 0, v = 0;
    auto tileL = [&](auto RR) {
      constexpr int R = decltype(RR)::value;
      static_for<NP>([&](auto PP) {
        constexpr int P = decltype(PP)::value;
        static_for<S::n_issue[P]>([&](auto E) {
          constexpr int e = decltype(E)::value;
          constexpr int u = S::issue_unit[P][e], d = S::issue_delta[P][e];
          if constexpr (R == 0 || d < R) {
            int stg = sidx + d;
            if (stg >= ST) stg -= ST;
            issue(IC<u>{}, d, stg);
          }
        });
        if constexpr (S::wait[R][P] >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EMU * S::wait[R][P]) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
      });
      ++v;
      va[0] = va[1]; vb[0] = vb[1]; va[1] = va[2]; vb[1] = vb[2];
      locate(v + 2, va[2], vb[2]);
      sidx = sidx + 1 == ST ? 0 : sidx + 1;
    };
    for (int t = 0; t < V - S::NTAIL; ++t) tileL(IC<0>{});
    static_for<S::NTAIL>([&](auto I) { tileL(IC<S::NTAIL - decltype(I)::value>{}); });
    __builtin_amdgcn_s_barrier();                                // group 0's cadence: one closing barrier
    return;
  }

  // -------------------------------------------------------------------------------------------------- compute waves
  const int wr = wave & 3, wc = wave >> 2;                       // wc = ping-pong group
  f32x4 acc[2][2][NF];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[WIDE ? 2 : 1][2][2], fb[3][2];
  const int rho = lane >> 4;
  constexpr int NQ = NF / 2;

  auto epilogue = [&](int jtile) {
    int tm, tn;
    tile_rc(jtile, tm, tn);
    const long m0 = (long)tm * 256, n0 = (long)tn * BN;
    float cs[NQ > 0 ? NQ : 1][8];
#pragma unroll
    for (int q = 0; q < (NQ > 0 ? NQ : 1); ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[q][e] = 0.f;
    float bias8[NQ > 0 ? NQ : 1][8];
#pragma unroll
    for (int q = 0; q < NQ; ++q) epi_bias8<EPI>(p, n0 + wc * WN + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8, bias8[q]);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int h = b >> 1, i = b & 1;
      const long row = m0 + wr * 64 + h * 32 + i * 16 + (lane & 15);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        float v8[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[h][i][2 * q][e]), __float_as_uint(acc[h][i][2 * q + 1][e]), false, false);
          v8[e] = __uint_as_float(r[0]); v8[4 + e] = __uint_as_float(r[1]);
        }
        const long col = n0 + wc * WN + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8;
        EpiIn8 in;
        epi_in8<EPI>(p, row, col, in);
        epi_out8<EPI>(p, v8, bias8[q], in, row, col);
        if (EPI == EPI_DGELU_BF16) {
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[q][e] += v8[e];
        }
      }
      if constexpr (NF & 1) epi_store<EPI>(p, acc[h][i][NF - 1], row, n0 + wc * WN + (NF - 1) * 16 + rho * 4);
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (EPI == EPI_DGELU_BF16 && p.colsum_part) {
      // per-128-row column sums (the FFN1 bias gradient): 16 lanes -> 1 by shuffles, then the two wave rows of a half through
      // float atomics into the zero-initialised output (two adders per address: order-independent up to one rounding)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = cs[q][e];
          t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
          if ((lane & 15) == 0)
            atomicAdd(p.colsum_part + ((long)tm * 2 + (wr >> 1)) * p.N + n0 + wc * WN + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8 + e, t);
        }
    }
  };

  __builtin_amdgcn_s_barrier();                                  // the prologue's barrier (its DMA was the loaders')
  if (wc == 1) __builtin_amdgcn_s_barrier();                     // group 1 runs one barrier behind group 0
  int sidx = 0, kt = 0, jtile = 0;
  auto tileC = [&]() {
    const char* st = smem + sidx * STAGE;
    static_for<NP>([&](auto PP) {
      constexpr int P = decltype(PP)::value;
      constexpr int h = S::phase_h[P], j = S::phase_j[P];
      if constexpr (WIDE) {
        if constexpr (P == 0) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int ks = 0; ks < 2; ++ks) fa[hh][i][ks] = frag16_row(st, wr * 64 + hh * 32 + i * 16, ks * 32);
        }
      } else if constexpr (P == 0 || S::phase_h[P] != S::phase_h[P == 0 ? 0 : P - 1]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) fa[0][i][ks] = frag16_row(st, wr * 64 + h * 32 + i * 16, ks * 32);
      }
      if constexpr (P == 0 || S::phase_j[P] != S::phase_j[P == 0 ? 0 : P - 1]) {
        const char* pb = st + PP_A_BYTES + j * BPART;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            fb[jj][ks] = BT ? frag16_col(pb, wc * 48 + jj * 16, ks * 32) : frag16_row(pb, wc * 48 + jj * 16, ks * 32);
      }
      if constexpr (WIDE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads complete before the barrier: war = 1
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(CAREL_PP_MPRIO);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hh = (WIDE ? 0 : h); hh < (WIDE ? 2 : h + 1); ++hh)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
              acc[hh][i][j * 3 + jj] = mfma16(fb[jj][ks], fa[WIDE ? hh : 0][i][ks], acc[hh][i][j * 3 + jj]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    });
    sidx = sidx + 1 == ST ? 0 : sidx + 1;
    if (++kt == nk) {                                            // tile complete: fire its stores, go on with the next tile
      epilogue(jtile);
      kt = 0; ++jtile;
    }
  };
  for (int t = 0; t < V; ++t) tileC();
  if (wc == 0) __builtin_amdgcn_s_barrier();
}

template <int NPN, bool BT, int EPI, int NLW, bool WIDE>
int launch_ppl(GemmParams p, hipStream_t s) {
  using G = PPGeom<NPN, BT>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_ppl_kernel<NPN, BT, EPI, NLW, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "gemm_ppl_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr = true;
  }
  p.tiles_m = p.M / 256; p.tiles_n = p.N / (96 * NPN);
  const int tiles = p.tiles_m * p.tiles_n;
  if (EPI == EPI_DGELU_BF16 && p.colsum_part) {                  // accumulated with atomics in the epilogue
    hipError_t e = hipMemsetAsync(p.colsum_part, 0, (size_t)(p.M / 128) * p.N * 4, s);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "gemm_ppl_kernel: memset: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL((gemm_ppl_kernel<NPN, BT, EPI, NLW, WIDE>), dim3(tiles < 256 ? tiles : 256), dim3(512 + 64 * NLW), G::LDS, s, p);
  return check_launch("gemm_ppl_kernel");
}

template <int NPN, bool AT, bool BT, int EPI, int DBG = 0, bool WIDE = false>
int launch_pp(GemmParams p, int splits, hipStream_t s) {
  using G = PPGeom<NPN, BT>;
  static bool attr = false;      // per process; setting it again is harmless if two threads race
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_pp_kernel<NPN, AT, BT, EPI, DBG, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "gemm_pp_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr = true;
  }
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = p.N / (96 * NPN);
  hipLaunchKernelGGL((gemm_pp_kernel<NPN, AT, BT, EPI, DBG, WIDE>), dim3(p.tiles_m * p.tiles_n, 1, splits), dim3(512), G::LDS, s, p);
  return check_launch("gemm_pp_kernel");
}

static int g_pp_wide = 0;        // tuning hook (carel_gemm_set_variant(90 / 91)): the wide-phase schedule where it is built (npn 2)

template <bool BT, int EPI>
int launch_pp_n(const GemmParams& p, int npn, hipStream_t s) {
  if (npn == 1) return g_pp_wide ? launch_pp<1, false, BT, EPI, 0, true>(p, 1, s) : launch_pp<1, false, BT, EPI>(p, 1, s);
  if (npn == 2) return g_pp_wide ? launch_pp<2, false, BT, EPI, 0, true>(p, 1, s) : launch_pp<2, false, BT, EPI>(p, 1, s);
  if constexpr (!BT) { if (npn == 3) return g_pp_wide ? launch_pp<3, false, BT, EPI, 0, true>(p, 1, s) : launch_pp<3, false, BT, EPI>(p, 1, s); }
  return set_error(CAREL_ERR_ARG, "gemm_pp_launch: npn = %d not built for this form", npn);
}

}  // namespace

void gemm_pp_wide_variant(int on) { g_pp_wide = on ? 1 : 0; }
static int g_pp_force_npn = 0;     // tuning hook (carel_gemm_set_variant(70 + n)): tile width 96 n wherever N allows; 0 = heuristic
void gemm_pp_force_npn(int n) { g_pp_force_npn = (n >= 1 && n <= 3) ? n : 0; }

// npn (1..3) when the ping-pong kernel should run this GEMM, 0 when it cannot or should not.
int gemm_pp_pick(const GemmParams& p, bool bt, int epi, int force) {
  if (p.N % 96 || p.K % 64 || p.K < 256 || p.M < 1) return 0;
  if (bt ? !(epi == EPI_BIAS_BF16 || epi == EPI_DGELU_BF16 || epi == EPI_ADD_F32)
         : !(epi == EPI_BIAS_BF16 || epi == EPI_BIAS_GELU || epi == EPI_BIAS_DROP_RESID || epi == EPI_ADD_F32)) return 0;
  const int tiles_m = (p.M + 255) / 256;
  int best = 0; double best_score = 0.0;
  for (int npn = bt ? 2 : 3; npn >= 1; --npn) {
    if (p.N % (96 * npn)) continue;
    if (g_pp_force_npn && npn != g_pp_force_npn && p.N % (96 * g_pp_force_npn) == 0 && !(bt && g_pp_force_npn == 3)) continue;
    if (epi == EPI_DGELU_BF16 && p.colsum_part && (npn & 1)) continue;      // the fused column sums need fragment pairs
    const long tiles = (long)tiles_m * (p.N / (96 * npn));
    const long rounds = (tiles + 255) / 256;
    const double fill = (double)tiles / (double)(rounds * 256);           // share of the CU-rounds that do work
    // staged bytes per FLOP fall with the tile width: (256 + 96 npn) / (256 * 96 npn)
    const double intensity = (256.0 * 96.0 * npn) / (256.0 + 96.0 * npn);
    const double score = fill * (intensity < 110.0 ? intensity : 110.0);  // past ~110 FLOP/B the loop is MFMA-bound
    if (score > best_score) { best_score = score; best = npn; }
  }
  if (!best) return 0;
  if (force <= 0) {                                                       // force = -(minimum tile count)
    const long tiles = (long)tiles_m * (p.N / (96 * best));
    if (tiles < (long)(-force)) return 0;                                 // small grids: the 128x128 kernel (+ split-K) fills the chip better
  }
  return best;
}

// Weight-gradient form (A^T B, K = tokens, fp32 slabs): npn for a split-K factor the caller has already fixed; 0 = cannot.
int gemm_pp_pick_tn(const GemmParams& p, int splits) {
  if (p.M % 256 || p.N % 96 || p.K % 64 || splits < 1) return 0;
  const int nk = p.K >> 6;
  if (nk / splits < 4) return 0;                                          // every slice needs a few K tiles (static schedule)
  int best = 0; double best_score = 0.0;
  for (int npn = 2; npn >= 1; --npn) {
    if (p.N % (96 * npn)) continue;
    const long wgs = (long)(p.M / 256) * (p.N / (96 * npn)) * splits;
    const long rounds = (wgs + 255) / 256;
    const double fill = (double)wgs / (double)(rounds * 256);
    const double intensity = (256.0 * 96.0 * npn) / (256.0 + 96.0 * npn);
    const double score = fill * (intensity < 110.0 ? intensity : 110.0);
    if (score > best_score) { best_score = score; best = npn; }
  }
  return best;
}

// The split-K factor the ping-pong kernel wants for dW[M,N] = A^T B over K tokens: as many slices as keep <= 256
// workgroups of the wider tile, each with at least 8 K tiles, at most 16 slabs.  0 = shape not supported.
int gemm_pp_wgrad_splits(int M, int N, long K) {
  if (M % 256 || N % 96 || K % 64 || K < 512) return 0;
  // measured at K = 8192 (tools/bench_gemm_pp.py, GEMM + slab reduction): 768 x 3072 / 3072 x 768: 64 us vs 82 us with the 128x128
  // kernel; 2304 x 768: 56.4 vs 53.9; 768 x 768: 33.5 vs 31.2 -- the small outputs need so many K slices to fill 256 CUs that the
  // slab traffic eats the gain, so they stay on the 128x128 kernel
  if ((long)M * N < 2000000L) return 0;
  const int npn = N % 192 == 0 ? 2 : 1;
  const long tiles = (long)(M / 256) * (N / (96 * npn));
  long s = 256 / tiles;
  const long nk = K >> 6;
  if (s > nk / 8) s = nk / 8;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  return (int)s;
}

int gemm_pp_launch_tn(const GemmParams& p, int npn, int splits, hipStream_t s) {
  if (npn == 1) return g_pp_wide ? launch_pp<1, true, true, EPI_SLAB_F32, 0, true>(p, splits, s) : launch_pp<1, true, true, EPI_SLAB_F32>(p, splits, s);
  if (npn == 2) return g_pp_wide ? launch_pp<2, true, true, EPI_SLAB_F32, 0, true>(p, splits, s) : launch_pp<2, true, true, EPI_SLAB_F32>(p, splits, s);
  return set_error(CAREL_ERR_ARG, "gemm_pp_launch_tn: npn = %d not built", npn);
}

#ifdef CAREL_GEMM_ABLATE
int gemm_pp_launch_dbg(const GemmParams& p, int npn, int dbg, hipStream_t s) {     // NT, bias -> bf16 epilogue only
#define PPD(N, D) if (npn == N && dbg == D) return g_pp_wide ? launch_pp<N, false, false, EPI_BIAS_BF16, D, true>(p, 1, s) : launch_pp<N, false, false, EPI_BIAS_BF16, D>(p, 1, s)
#define PPDN(N) PPD(N, 1); PPD(N, 2); PPD(N, 3); PPD(N, 4); PPD(N, 5); PPD(N, 6); PPD(N, 7); PPD(N, 8)
  PPDN(1); PPDN(2); PPDN(3);
#undef PPDN
#undef PPD
  return set_error(CAREL_ERR_ARG, "gemm_pp_launch_dbg: (npn, dbg) = (%d, %d) not built", npn, dbg);
}
#endif

static int g_pp_loader = 0;      // tuning hook (carel_gemm_set_variant(80 / 81)): the loader-wave persistent variant off / on
void gemm_pp_loader_variant(int on) { g_pp_loader = (on >= 0 && on <= 2) ? on : 0; }

// mode 1: two loader waves, fine schedule (the first experiment); mode 2: four loader waves -- with the wide schedule for npn 1
// (12 waves = 3 per SIMD leave 170 registers per wave: the wide npn-2 kernel's 190-200 do not fit), fine for npn 2
template <bool BT, int EPI>
static int launch_ppl_n(const GemmParams& p, int npn, hipStream_t s) {
  if (g_pp_loader == 2) {
    if (npn == 1) return launch_ppl<1, BT, EPI, 4, true>(p, s);
    return launch_ppl<2, BT, EPI, 4, false>(p, s);
  }
  if (npn == 1) return launch_ppl<1, BT, EPI, 2, false>(p, s);
  return launch_ppl<2, BT, EPI, 2, false>(p, s);
}

int gemm_pp_launch(const GemmParams& p, bool bt, int epi, int npn, hipStream_t s) {
  const int nk = p.K >> 6;
  if (g_pp_loader && npn <= 2 && p.M % 256 == 0 && nk >= 4) {
    const int tiles = (p.M / 256) * (p.N / (96 * npn));
    if (tiles <= 256 || tiles % 8 == 0) {
      if (!bt) {
        switch (epi) {
          case EPI_BIAS_BF16: return launch_ppl_n<false, EPI_BIAS_BF16>(p, npn, s);
          case EPI_BIAS_GELU: return launch_ppl_n<false, EPI_BIAS_GELU>(p, npn, s);
          case EPI_BIAS_DROP_RESID: return launch_ppl_n<false, EPI_BIAS_DROP_RESID>(p, npn, s);
          case EPI_ADD_F32: return launch_ppl_n<false, EPI_ADD_F32>(p, npn, s);
        }
      } else {
        switch (epi) {
          case EPI_BIAS_BF16: return launch_ppl_n<true, EPI_BIAS_BF16>(p, npn, s);
          case EPI_DGELU_BF16: return launch_ppl_n<true, EPI_DGELU_BF16>(p, npn, s);
          case EPI_ADD_F32: return launch_ppl_n<true, EPI_ADD_F32>(p, npn, s);
        }
      }
    }
  }
  if (!bt) {
    switch (epi) {
      case EPI_BIAS_BF16: return launch_pp_n<false, EPI_BIAS_BF16>(p, npn, s);
      case EPI_BIAS_GELU: return launch_pp_n<false, EPI_BIAS_GELU>(p, npn, s);
      case EPI_BIAS_DROP_RESID: return launch_pp_n<false, EPI_BIAS_DROP_RESID>(p, npn, s);
      case EPI_ADD_F32: return launch_pp_n<false, EPI_ADD_F32>(p, npn, s);
    }
  } else {
    switch (epi) {
      case EPI_BIAS_BF16: return launch_pp_n<true, EPI_BIAS_BF16>(p, npn, s);
      case EPI_DGELU_BF16: return launch_pp_n<true, EPI_DGELU_BF16>(p, npn, s);
      case EPI_ADD_F32: return launch_pp_n<true, EPI_ADD_F32>(p, npn, s);
    }
  }
  return set_error(CAREL_ERR_ARG, "gemm_pp_launch: unsupported form/epilogue (%d,%d)", (int)bt, epi);
}

}  // namespace carel
