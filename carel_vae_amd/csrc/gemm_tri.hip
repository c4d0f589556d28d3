// Three-group bf16 MFMA GEMM for the N = 768 forward linears of the encoder (row-major A, nn.Linear-layout weight: the NT form):
// replaces nn.Linear.forward inside HF BertSelfOutput / BertOutput (drl_classifier_ec_mmd_final_mul.py:202-206).
// EXPERIMENT (tuning hook 221; OFF by default): bit-identical to the ping-pong kernel (same order of additions; tests/test_gpu_gemm.py)
// and NOT faster -- FFN2 forward 48.9 us against 41.6 on one box, out-projection 17.2 against 15.8.  Its own ablations
// (tools/ablate_gemm_tri.py: no DMA 37.0, no MFMA 38.8, no fragment reads 35.8, barriers only 10.1) show the ping-pong kernel's
// picture once more: any two of {LDS-DMA, fragment reads, MFMA} overlap, all three do not, whatever the wave schedule.  What the three
// share is the CU's vector-memory return path and the LDS behind it: 44 KB per K tile at the ~56-64 B/clk a CU takes in
// (tools/ubench/ingest.hip) are 700-800 cycles against 768 of MFMA issue -- at 70 FLOP per staged byte the 256 x 96 tile is
// co-limited by both, and N = 768 at T = 8192 admits no wider tile on 256 CUs.
//
// Why: in the two-group ping-pong kernel (gemm_pp.hip) a wave's load segment (14 fragment reads + its LDS-DMA issue + waits: 400-480
// cycles) is as long as the 24-MFMA segment it hides behind (384 alone, 450-520 beside the partner's reads), so a K tile costs
// ~1 330 cycles against 768 of MFMA issue.  Here the same 256 x 96 x 64 tile is worked by TWELVE waves = 4 (M) x 3 (N), wave tile
// 64 x 32 (16 MFMAs 16x16x32 per K tile), three waves per SIMD: the wave columns are three groups that rotate through
//     L1 (A fragments: 8 ds_read_b128)  ->  L2 (B fragments: 4 ds_read_b128)  ->  M (16 MFMAs)
// one slot apart, one raw s_barrier per slot: while one group multiplies, the other two load, and a load may take two slots.
//
// Absolute slots: group g runs L1(j) in slot 3j + g, L2(j) in 3j + g + 1, M(j) in 3j + g + 2 for K tile j.  Stage j % 3 is read in
// slots 3j .. 3j + 3 and free from 3j + 4; tile j + 3 is first read in slot 3j + 9.  Every wave copies 4 pieces (1 KiB each) of every
// tile (48 slots for 44 pieces: the last four repeat a piece -- same bytes to the same place):
//     group 0 issues tile i + 2 in L2(i) (slot 3i + 1 >= 3(i - 1) + 4) and waits for tile i + 1 at the end of M(i);
//     group 1 issues tile i + 2 in L1(i) (slot 3i + 1)                  and waits for tile i + 1 at the end of L2(i);
//     group 2 issues tile i + 2 in L1(i) (slot 3i + 2)                  and waits for tile i + 1 at the end of L1(i);
// all three waits fall in slot 3i + 2, whose barrier publishes tile i + 1 before its first read (slot 3i + 3), and each leaves exactly
// the 4 newer pieces (tile i + 2) in flight: s_waitcnt vmcnt(4), vmcnt(0) once nothing newer was issued.
#ifdef CAREL_EXPERIMENTS      // an experiment (built, measured, not adopted): not part of the product library
#include "gemm_epilogue.h"

namespace carel {
namespace {

constexpr int TRI_A = 32768, TRI_B = 12288, TRI_STAGE = TRI_A + TRI_B, TRI_LDS = 3 * TRI_STAGE;

// DBG (timing ablations, wrong results; hooks 222-225): 1 no LDS-DMA after the prologue, 2 no MFMA, 3 no fragment reads after the first tile, 4 barriers only
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(768) void gemm_tri_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave & 3, wc = wave >> 2;                    // wc = group: waves w, w + 4, w + 8 share a SIMD
  // XCD-aware tile map: each XCD walks a contiguous chunk of the row-major tile order
  int tm, tn;
  {
    const int nwg = p.tiles_m * p.tiles_n, flat = (int)blockIdx.x;
    const int xcd = flat & 7, qq = nwg >> 3, rr = nwg & 7;
    const int item = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (flat >> 3);
    tm = item / p.tiles_n; tn = item - tm * p.tiles_n;
  }
  const long m0 = (long)tm * 256, n0 = (long)tn * 96;
  // ---- this wave's four pieces: per-lane source offsets (constant over K) and LDS destinations
  uint32_t soff[4];
  int sdst[4];
  bool isb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int pc = wave + 12 * i;                                   // 0..47
    if (pc >= 44) pc -= 12;                                   // the four spare slots repeat pieces 32..35 (B rows 0..31)
    const int r8 = lane >> 3, c = (lane & 7) ^ r8;            // source chunk = physical chunk ^ (row & 7)
    if (pc < 32) {
      long rg = m0 + pc * 8 + r8; if (rg > (long)p.M - 1) rg = (long)p.M - 1;
      soff[i] = (uint32_t)(((rg - m0) * p.lda + c * 8) * 2);
      sdst[i] = pc * 1024; isb[i] = false;
    } else {
      const int row = (pc - 32) * 8 + r8;
      soff[i] = (uint32_t)(((long)row * p.ldb + c * 8) * 2);
      sdst[i] = TRI_A + (pc - 32) * 1024; isb[i] = true;
    }
  }
  const int nk = p.K >> 6;
  const char* a_base = (const char*)(p.A + m0 * p.lda);
  const char* b_base = (const char*)(p.B + n0 * p.ldb);
  auto issue = [&](int tile) {                                // this wave's 4 pieces of K tile `tile` into stage tile % 3
    char* sb = smem + (tile % 3) * TRI_STAGE;
    const long koff = (long)tile * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* g = (isb[i] ? b_base : a_base) + koff + soff[i];
      __builtin_amdgcn_global_load_lds((const void*)g, (CAREL_LDS void*)(sb + sdst[i]), 16, 0, 0);
      asm volatile("" ::: "memory");
    }
  };
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4][2], fb[2][2];

  issue(0); issue(1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");            // tile 0 landed (this wave's share)
  __builtin_amdgcn_s_barrier();
  for (int g = 0; g < wc; ++g) __builtin_amdgcn_s_barrier();  // group g starts g slots late

  for (int i = 0; i < nk; ++i) {
    const char* st = smem + (i % 3) * TRI_STAGE;
    const bool more = i + 2 < nk;                             // wave-uniform
    // ---------------- L1(i): A fragments
    const bool dma = DBG != 1 && DBG != 4, reads = (DBG != 3 && DBG != 4) || i == 0;
    if (wc != 0 && more && dma) issue(i + 2);
    if (reads)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[b][ks] = frag16_row(st, wr * 64 + b * 16, ks * 32);
    if (wc == 2 && dma) {
      if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- L2(i): B fragments
    if (wc == 0 && more && dma) issue(i + 2);
    if (reads)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb[j][ks] = frag16_row(st + TRI_A, wc * 32 + j * 16, ks * 32);
    if (wc == 1 && dma) {
      if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- M(i)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (DBG == 2 || DBG == 4) asm volatile("" ::"v"(fb[j][ks]), "v"(fa[b][ks]));
          else acc[b][j] = mfma16(fb[j][ks], fa[b][ks], acc[b][j]);     // swapped: D[n][m]
        }
    __builtin_amdgcn_s_setprio(0);
    if (wc == 0 && dma) {
      if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int g = wc; g < 2; ++g) __builtin_amdgcn_s_barrier();  // every wave has now passed the same number of barriers

  // ---- epilogue straight from the accumulators (the ping-pong kernel's scheme): v_permlane16_swap pairs the two 16-column fragments
  // so that every lane owns 8 consecutive columns of one row
  const int rho = lane >> 4;
  const long col = n0 + wc * 32 + (rho & 1) * 16 + (rho >> 1) * 8;
  float bias8[8], ln8[16];
  epi_bias8<EPI>(p, col, bias8);
  epi_ln8<EPI>(p, col, ln8);
  EpiIn8 in[2];
  auto row_of = [&](int b) { return m0 + wr * 64 + b * 16 + (lane & 15); };
  if (row_of(0) < (long)p.M) epi_in8<EPI>(p, row_of(0), col, in[0]);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (b + 1 < 4 && row_of(b + 1) < (long)p.M) epi_in8<EPI>(p, row_of(b + 1), col, in[(b + 1) & 1]);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[b][0][e]), __float_as_uint(acc[b][1][e]), false, false);
      v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
    }
    const long row = row_of(b);
    if (row < (long)p.M) epi_out8<EPI>(p, v, bias8, in[b & 1], row, col, nullptr, ln8);
  }
}

template <int EPI, int DBG = 0>
int launch_tri(GemmParams p, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tri_kernel<EPI, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, TRI_LDS);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "gemm_tri_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr = true;
  }
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = p.N / 96;
  hipLaunchKernelGGL((gemm_tri_kernel<EPI, DBG>), dim3(p.tiles_m * p.tiles_n), dim3(768), TRI_LDS, s, p);
  return check_launch("gemm_tri_kernel");
}

}  // namespace

static int g_tri = 0;              // tuning hook (carel_gemm_set_variant(220 / 221; 222-225 with -DCAREL_GEMM_ABLATE: timing ablations 1-4))
void gemm_tri_enable(int on) { g_tri = on; }
// 1 when the three-group kernel should take this NT GEMM (experiment: only when switched on)
int gemm_tri_pick(const GemmParams& p, int epi) {
  if (!g_tri || p.N % 96 || p.K % 64 || p.K < 256 || p.M < 1 || p.colsum_part) return 0;
  if (!(epi == EPI_BIAS_BF16 || epi == EPI_BIAS_DROP_RESID || epi == EPI_ADD_F32)) return 0;
  const long tiles = (long)((p.M + 255) / 256) * (p.N / 96);
  return tiles >= 192 && tiles <= 256;
}
int gemm_tri_launch(const GemmParams& p, int epi, hipStream_t s) {
#ifdef CAREL_GEMM_ABLATE
  if (epi == EPI_BIAS_BF16 && g_tri == 2) return launch_tri<EPI_BIAS_BF16, 1>(p, s);
  if (epi == EPI_BIAS_BF16 && g_tri == 3) return launch_tri<EPI_BIAS_BF16, 2>(p, s);
  if (epi == EPI_BIAS_BF16 && g_tri == 4) return launch_tri<EPI_BIAS_BF16, 3>(p, s);
  if (epi == EPI_BIAS_BF16 && g_tri == 5) return launch_tri<EPI_BIAS_BF16, 4>(p, s);
#endif
  switch (epi) {
    case EPI_BIAS_BF16: return launch_tri<EPI_BIAS_BF16>(p, s);
    case EPI_BIAS_DROP_RESID: return launch_tri<EPI_BIAS_DROP_RESID>(p, s);
    case EPI_ADD_F32: return launch_tri<EPI_ADD_F32>(p, s);
  }
  return set_error(CAREL_ERR_ARG, "gemm_tri_launch: unsupported epilogue %d", epi);
}

}  // namespace carel

#endif   // CAREL_EXPERIMENTS
