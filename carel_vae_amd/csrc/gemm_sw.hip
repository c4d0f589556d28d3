// One-wave-per-SIMD bf16 MFMA GEMM for the 768-wide outputs of the encoder at T = 8192 (row-major-A forms: forward NT, data gradient NN):
// replaces nn.Linear.forward / the dX half of its backward inside HF BertSelfOutput / BertOutput / BertIntermediate / BertSelfAttention
// (drl_classifier_ec_mmd_final_mul.py:202-206, :841) where the output is 768 wide -- out-projection and FFN2 forward, FFN1 / QKV /
// out-projection data gradients: five of the eight GEMMs of a layer.
//
// Why a third main loop.  The ping-pong kernel (gemm_pp.hip) runs these shapes as 256 tiles of 256 x 96 at 52 % of the MFMA rate: two
// wave groups alternate a load segment and a 24-MFMA segment between barriers, a K tile costs two slots of max(load, MFMA) + barrier
// = 1 300-1 460 cycles against 768 of MFMA issue, and round 3 showed that neither fewer LDS reads (K split inside the workgroup), nor
// fewer staged bytes (pair split-K with 256 x 192 tiles), nor reading fragments ahead (the groups' one-barrier skew leaves no legal
// issue window with three stages) moves it.  Here the SAME tile is computed by FOUR waves, one per SIMD, each 64 rows x 96 columns
// (4 x 6 accumulators, 96 registers), all in lockstep:
//   * one workgroup barrier per 64-deep K tile, in the middle of its 48 MFMAs;
//   * fragments double-buffered in registers per 32-deep half: while the 24 MFMAs of one half run, the 4 + 6 ds_read_b128 of the next
//     half (next tile after the barrier) are in flight -- the wave never waits for a fragment it has just requested;
//   * three LDS stages, LDS-DMA two tiles ahead, ONE counted s_waitcnt vmcnt per tile (every tile is 11 DMA instructions per wave, so
//     the immediate is static: 11 in the steady state, 0 for the last landing tile);
//   * 80 KiB of fragment reads + 44 KiB of DMA per K tile through the LDS (the ping-pong tiling: 112 + 44);
//   * the epilogue is the ping-pong kernel's (8 consecutive columns per lane after v_permlane16_swap; the inputs of all four row blocks
//     requested behind the prologue's DMA and held through the main loop -- with one wave per SIMD there are 512 registers).
// Same accumulation order per output as the other kernels (K tiles in order, k32 halves in order, one chain): the same bits as
// gemm_pp_kernel / gemm_kernel (tests/test_gpu_gemm.py).
#include <utility>
#include "gemm_epilogue.h"

namespace carel {
namespace {

template <int V> struct SIC { static constexpr int value = V; };
template <class F, int... I>
__device__ __forceinline__ void sw_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(SIC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sw_static_for(F&& f) { sw_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int SW_A_BYTES = 32768, SW_ST = 2, SW_D = 2;       // two LDS stages; SW_D K tiles in flight in registers behind the one being written
template <bool BT> struct SWGeom {
  static constexpr int BPART = BT ? 16384 : 12288;
  static constexpr int STAGE = SW_A_BYTES + BPART;
  static constexpr int LDS = SW_ST * STAGE;
};

// DBG (timing only, wrong results): 1 = no MFMA, 2 = no epilogue
template <bool BT, int EPI, int DBG>
__global__ __launch_bounds__(256, 1) void gemm_sw_kernel(GemmParams p) {
  using G = SWGeom<BT>;
  constexpr int STAGE = G::STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // row band: rows wave * 64 .. + 64 of the tile
  // XCD-aware tile map (the ping-pong kernel's rectangles: gemm_pp.hip)
  int tm, tn;
  {
    const int tiles = p.tiles_m * p.tiles_n, flat = (int)blockIdx.x;
    const int xcd = flat & 7, qq = tiles >> 3, rr = tiles & 7;
    const int item = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (flat >> 3);
    if (p.pp_xr) {
      const int xc = 8 / p.pp_xr, R = p.tiles_m / p.pp_xr, C = p.tiles_n / xc, local = flat >> 3;
      const int xi = xcd / xc, xj = xcd - xi * xc;
      const int per = R * p.pp_bc, blk = local / per, rem = local - blk * per;
      const int r = rem / p.pp_bc;
      tm = xi * R + r; tn = xj * C + blk * p.pp_bc + (rem - r * p.pp_bc);
    } else {
      tm = item / p.tiles_n; tn = item - tm * p.tiles_n;
    }
  }
  const long m0 = (long)tm * 256, n0 = (long)tn * 96;
  const int nk = p.K >> 6;

  // ---- staging: global -> registers (SW_D tiles deep) -> LDS.  Per thread and K tile: 8 chunks of 16 B of the A tile (256 rows x 128 B, a
  // wave instruction covers 8 whole rows) and 3 of the B tile (NT: 96 weight rows x 128 B; NN: 64 k-rows x 192 B).  The LDS images are
  // the ROW / COL images of carel_common.h (XOR swizzle on the write address).
  uint32_t aoff[8]; int adst[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = i * 256 + tid, row = c >> 3, ch = c & 7;
    long rg = m0 + row; if (rg > (long)p.M - 1) rg = (long)p.M - 1;             // rows past M re-read the last row (never stored)
    aoff[i] = (uint32_t)(((rg - m0) * p.lda + ch * 8) * 2);
    adst[i] = row_img_off(row, ch);
  }
  uint32_t boff[3]; int bdst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + tid;
    if (!BT) { const int r = c >> 3, ch = c & 7; boff[i] = (uint32_t)(((long)r * p.ldb + ch * 8) * 2); bdst[i] = SW_A_BYTES + row_img_off(r, ch); }
    else { const int r = c / 12, ch = c - r * 12; boff[i] = (uint32_t)(((long)r * p.ldb + ch * 8) * 2); bdst[i] = SW_A_BYTES + col_img_off(r, ch); }
  }
  const long a_step = 128, b_step = BT ? 64 * p.ldb * 2 : 128;
  const char* ga = (const char*)(p.A + m0 * p.lda);            // wave-uniform pointers of the next K tile to request
  const char* gb = (const char*)(BT ? p.B + n0 : p.B + n0 * p.ldb);
  bf16x8 stg[SW_D][11];                                      // (an ext_vector type: arrays of HIP's uint4 class stay in scratch memory)
  auto request = [&](auto SET) {                               // the next K tile -> register set SET
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < 8; ++i) stg[set][i] = *(const bf16x8*)(ga + aoff[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) stg[set][8 + i] = *(const bf16x8*)(gb + boff[i]);
    ga += a_step; gb += b_step;
  };
  auto deposit = [&](auto SET, int stage_off) {                // register set SET -> LDS stage
    constexpr int set = decltype(SET)::value;
    char* sb = smem + stage_off;
#pragma unroll
    for (int i = 0; i < 8; ++i) *(bf16x8*)(sb + adst[i]) = stg[set][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) *(bf16x8*)(sb + bdst[i]) = stg[set][8 + i];
  };

  f32x4 acc[4][6];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][4], fb[2][6];
  auto read_frags = [&](auto SET, int stage_off, int half) {
    constexpr int set = decltype(SET)::value;
    const char* st = smem + stage_off;
#pragma unroll
    for (int r = 0; r < 4; ++r) fa[set][r] = frag16_row(st, wave * 64 + r * 16, half * 32);
#pragma unroll
    for (int j = 0; j < 6; ++j) fb[set][j] = BT ? frag16_col(st + SW_A_BYTES, j * 16, half * 32) : frag16_row(st + SW_A_BYTES, j * 16, half * 32);
  };
  auto mma = [&](auto SET) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (DBG == 1) asm volatile("" ::"v"(fb[set][j]), "v"(fa[set][r]));
        else acc[r][j] = mfma16(fb[set][j], fa[set][r], acc[r][j]);              // swapped operands: D[n][m]
      }
  };

  // ---- epilogue geometry and its inputs, requested right behind the first tiles (plain loads: the compiler's in-order vmcnt counts them) ----
  const int rho = lane >> 4;
  auto row_of = [&](int b) { return m0 + wave * 64 + b * 16 + (lane & 15); };
  auto col_of = [&](int q) { return n0 + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8; };
  constexpr bool HAS_IN = EPI == EPI_BIAS_DROP_RESID || EPI == EPI_ADD_F32 || epi_is_dgelu(EPI);
  EpiIn8 pin[4][3];

  // ---- prologue: tiles 0 .. SW_D requested, tile 0 deposited -----------------------------------------------------------------------------
  request(SIC<0>{});
  request(SIC<1>{});
  deposit(SIC<0>{}, 0);
  request(SIC<0>{});                                           // tile 2 (nk >= 4: host check)
  __syncthreads();
  read_frags(SIC<0>{}, 0, 0);

  // One K tile t (its fragments of the first half are in registers).  SET: the register set holding tile t + 1 (deposited here, then re-used
  // for tile t + 1 + SW_D).  MORE: 2 = request another tile, 1 = only deposit tile t + 1, 0 = last tile.
  int s_cur = 0;
  auto tile = [&](auto SETC, auto MOREC) {
    constexpr int SET = decltype(SETC)::value, MORE = decltype(MOREC)::value;
    const int s_next = s_cur ^ STAGE;                          // two stages: offsets 0 and STAGE
    read_frags(SIC<1>{}, s_cur, 1);
    if constexpr (MORE >= 1) deposit(SIC<SET>{}, s_next);      // stage of tile t - 1: every wave finished reading it before the last barrier
    if constexpr (MORE >= 2) request(SIC<SET>{});
    mma(SIC<0>{});
    __syncthreads();                                           // tile t + 1 is in the LDS for every wave; this tile's fragments are all in registers
    if constexpr (MORE >= 1) read_frags(SIC<0>{}, s_next, 0);
    mma(SIC<1>{});
    s_cur = s_next;
  };
  // tiles 0 .. nk-1; the set of tile t + 1 is (t + 1) % SW_D; tiles up to nk - 2 - SW_D still request (tile t + 1 + SW_D <= nk - 1)
  // nk is even and >= 4 (K a multiple of 128: host check), SW_D = 2: tiles 0 .. nk - 4 request another tile -- an odd count -- so one tile,
  // then pairs, then the three tail tiles: a static sequence, no dispatch on t inside the loop
  static_assert(SW_D == 2, "the tile sequence below is written for two register sets");
  tile(SIC<1>{}, SIC<2>{});
  for (int i = 0; i < (nk - 4) / 2; ++i) { tile(SIC<0>{}, SIC<2>{}); tile(SIC<1>{}, SIC<2>{}); }
  tile(SIC<0>{}, SIC<1>{});
  tile(SIC<1>{}, SIC<1>{});
  tile(SIC<0>{}, SIC<0>{});
  if (DBG == 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 6; ++j) asm volatile("" ::"v"(acc[r][j]));
    return;
  }

  // ---- epilogue: 4 row blocks x 3 column pairs, 8 consecutive columns per lane -----------------------------------------------------------
  if (HAS_IN) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      long row = row_of(b); if (row > (long)p.M - 1) row = (long)p.M - 1;
#pragma unroll
      for (int q = 0; q < 3; ++q) epi_in8<EPI>(p, row, col_of(q), pin[b][q]);
    }
  }
  float bias8[3][8];
#pragma unroll
  for (int q = 0; q < 3; ++q) epi_bias8<EPI>(p, col_of(q), bias8[q]);
  sw_static_for<4>([&](auto BB) {
    constexpr int b = decltype(BB)::value;
    const long row = row_of(b);
    const bool ok = row < (long)p.M;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[b][2 * q][e]), __float_as_uint(acc[b][2 * q + 1][e]), false, false);
        v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
      }
      if (ok) epi_out8<EPI>(p, v, bias8[q], pin[b][q], row, col_of(q), nullptr);
    }
  });
}

template <bool BT, int EPI, int DBG = 0>
int launch_sw(GemmParams p, hipStream_t s) {
  using G = SWGeom<BT>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_sw_kernel<BT, EPI, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "gemm_sw_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr = true;
  }
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = p.N / 96;
  p.pp_xr = 0; p.pp_bc = 1;
  long best = -1;                                              // XCD rectangles, as launch_pp
  for (int xr = 8; xr >= 1; xr >>= 1) {
    const int xc = 8 / xr;
    if (p.tiles_m % xr || p.tiles_n % xc) continue;
    const long cost = (long)(p.tiles_m / xr) * 256 + (long)(p.tiles_n / xc) * 96;
    if (best < 0 || cost < best) { best = cost; p.pp_xr = xr; }
  }
  if (p.pp_xr) {
    const int R = p.tiles_m / p.pp_xr, C = p.tiles_n / (8 / p.pp_xr);
    int bc = 1;
    for (int d = 1; d <= C; ++d) if (C % d == 0 && (long)R * d <= 32) bc = d;
    p.pp_bc = bc;
  }
  hipLaunchKernelGGL((gemm_sw_kernel<BT, EPI, DBG>), dim3(p.tiles_m * p.tiles_n), dim3(256), G::LDS, s, p);
  return check_launch("gemm_sw_kernel");
}

}  // namespace

static int g_sw_mode = 0;          // tuning hook (carel_gemm_set_variant(210 + m)): 0 = off, 1 = the 768-wide GEMMs whose grid fills the chip, 2 = + ablation (no MFMA)
void gemm_sw_mode(int m) { g_sw_mode = m; }

// 1 when this problem should run on the one-wave-per-SIMD kernel
int gemm_sw_pick(const GemmParams& p, bool bt, int epi) {
  if (!g_sw_mode) return 0;
  if (p.N % 96 || p.K % 128 || p.K < 256 || p.M < 1) return 0;
  if (bt ? !(epi == EPI_BIAS_BF16 || epi == EPI_ADD_F32) : !(epi == EPI_BIAS_BF16 || epi == EPI_BIAS_DROP_RESID || epi == EPI_ADD_F32)) return 0;
  if (p.colsum_part) return 0;
  const long tiles = (long)((p.M + 255) / 256) * (p.N / 96);
  return p.N == 768 && tiles >= 192 && tiles <= 256;
}
int gemm_sw_launch(const GemmParams& p, bool bt, int epi, hipStream_t s) {
  if (!bt) {
    switch (epi) {
      case EPI_BIAS_BF16: return launch_sw<false, EPI_BIAS_BF16>(p, s);
      case EPI_BIAS_DROP_RESID: return g_sw_mode == 2 ? launch_sw<false, EPI_BIAS_DROP_RESID, 1>(p, s) : launch_sw<false, EPI_BIAS_DROP_RESID>(p, s);   // (2: no MFMA -- timing only)
      case EPI_ADD_F32: return launch_sw<false, EPI_ADD_F32>(p, s);
    }
  } else {
    switch (epi) {
      case EPI_BIAS_BF16: return launch_sw<true, EPI_BIAS_BF16>(p, s);
      case EPI_ADD_F32: return launch_sw<true, EPI_ADD_F32>(p, s);
    }
  }
  return set_error(CAREL_ERR_ARG, "gemm_sw_launch: unsupported form/epilogue (%d,%d)", (int)bt, epi);
}

}  // namespace carel
