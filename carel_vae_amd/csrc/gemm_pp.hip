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
#include <atomic>
#include <type_traits>
#include <utility>
#include "gemm_epilogue.h"
#include "reduce_device.h"

namespace carel {

#include "gemm_pp_sched.inc"

namespace {

// the GELU table (gemm_epilogue.h: gelu_lut8): word i = bf16 gelu(u) | bf16 gelu'(u) << 16 for the i-th bf16 u of the covered range
__device__ uint32_t g_gelu_lut[LUT_WORDS];
__global__ void gelu_lut_fill_kernel() {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= LUT_WORDS) return;
  const uint32_t h = (uint32_t)((i % LUT_HALF) + (LUT_EXP_LO << 7)) | (i >= LUT_HALF ? 0x8000u : 0u);
  float u[8], d[8];
  unpack8(uint4{h | (h << 16), h | (h << 16), h | (h << 16), h | (h << 16)}, u);
#pragma unroll
  for (int e = 0; e < 8; e += 2) {             // the arithmetic path of epi_out8, verbatim
    f32x2 g, dg;
    gelu_erf_both2(f32x2{u[e], u[e + 1]}, g, dg);
    u[e] = g.x; u[e + 1] = g.y; d[e] = dg.x; d[e + 1] = dg.y;
  }
  g_gelu_lut[i] = (pack8(u).x & 0xffffu) | (pack8(d).x << 16);
}
// Filled by carel_init(device), once per device per process (capi.hip calls gemm_pp_init_device): the fill is enqueued on the null
// stream and carel_init synchronises it -- initialisation may; no GEMM call ever synchronises.  A GELU-epilogue launch on a device
// that was never initialised fails loudly instead of filling lazily (a lazy fill would need a device-wide synchronise inside a call).
static std::atomic<int> g_lut_state[16];       // 0 = never filled, 1 = filled
static int gelu_lut_ready() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return set_error(CAREL_ERR_HIP, "gemm_pp: hipGetDevice failed");
  if (g_lut_state[dev].load(std::memory_order_acquire) == 1) return CAREL_OK;
  return set_error(CAREL_ERR_ARG, "gemm_pp: carel_init(%d) has not been called on this device (it fills the GELU table of the fused FFN1 epilogue)", dev);
}

template <int V> struct IC { static constexpr int value = V; };
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

CAREL_TUNABLE(int, g_pp_gelu_lut, 1);    // tuning hook (carel_gemm_set_variant(160 / 161)): GELU epilogues by erf / exp arithmetic / by table lookup
CAREL_TUNABLE(int, g_pp_epi_prefetch, 1);   // tuning hook (carel_gemm_set_variant(170 / 171)): epilogue inputs requested at the end / before the main loop
CAREL_TUNABLE(int, g_pp_xcd_rect, 1);    // tuning hook (carel_gemm_set_variant(120 / 121)): XCD tile map of the NT / NN forms: row-major chunks / rectangles
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
// The first ten arguments repeat the fields of `p` that the prologue needs before its first LDS-DMA instruction: as plain scalar kernel
// arguments they are PRELOADED into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count=16, carel_vae_amd/build.py; a by-value
// struct is not), so the address arithmetic does not wait for the first s_load round trip of a cold CU.
// PAIR (pair split-K; NT / NN, gridDim.z == 2): the two workgroups of a tile take half of K each.  Slice 0 ("A") stores its raw
// accumulators -- [tile][wave][register][lane], 1 KiB per instruction -- and then a per-wave flag (this launch's sequence number), and
// exits; slice 1 ("B") waits for the flag of ITS wave index, adds A's partial sums to its own and runs the epilogue.  Why: the N = 768
// outputs are 256 tiles of 256 x 96 at T = 8192 -- 70 FLOP per staged byte, L2 -> LDS-bound and LDS-bandwidth-bound;
// as 128 tiles of 256 x 192 split in two along K they stage 36 % fewer bytes per FLOP and read 30 % fewer LDS bytes per FLOP, on all
// 256 CUs.  Deadlock-free by construction: the wait is one-directional (A never waits) and workgroups are dispatched in order of
// their flat index, x fastest, z slowest -- every A is on a CU (or finished) before the first B exists -- so it holds with other
// kernels or other processes sharing the GPU.  All pair traffic is system-scope (write-through stores, cache-bypassing loads): it
// does not rely on A and B sharing an XCD's L2.  The sum is (first half of K) + (second half): deterministic, but not the bits of the
// one-workgroup order.
// GROUP (weight-gradient form only; round 4): ONE launch computes up to four weight gradients dW_g[M_g, N_g] = dY_g^T X_g over the same T
// tokens -- the four linears of an encoder layer -- from a work list in the kernel arguments.  An item is a 256 x 96 output tile with either
// the WHOLE contraction (its result goes straight into dW: no slab, no reduction pass) or one of `s` K slices of a tile (a compact partial
// tile in the workspace, summed in slice order by wgrad_group_reduce_kernel).  The list holds a multiple of the CU count of whole tiles
// first and splits only the remainder (an encoder layer at T = 8192: 288 tiles = 256 whole + 32 x 8 slices; every CU runs 128 + 16 K
// tiles), so the slab traffic of the per-GEMM split-K launches (5 / 5 / 3 / 9 slabs written and read back: 1.98 GB per step) shrinks to
// the 32 split tiles (25 MB per layer) and four launches + four reductions become one + one.  Item order = dispatch order: workgroup i runs
// on XCD i % 8, and the list gives each XCD a run of consecutive tiles of ONE problem (same dY columns or same X columns: shared through
// its L2).  No workgroup waits for another: nothing here can deadlock, whatever else shares the GPU.
struct PPGroupProb { const bf16_t* A; const bf16_t* B; float* out; float* colsum; int M, N; };      // A = dY [T, M], B = X [T, N], out = dW [M, N], colsum = db [M] or null
constexpr int PP_GROUP_MAX_ITEMS = 768;
struct PPGroup {
  PPGroupProb prob[4];
  float* part;                       // [n_items - n_full][256 * 96] partial tiles of the split items, in item order
  float* cs_part;                    // [n_items - n_full][256] their bias-gradient partials
  int nk, s, n_full, n_items;        // K tiles (T / 64); slices per split tile; whole-tile items; all items
  unsigned short item[PP_GROUP_MAX_ITEMS];    // problem | tile row << 2 | tile column << 6 | (slice + 1, 0 = whole) << 11
};
struct PPNoGroup {};

template <int NPN, bool AT, bool BT, int EPI, int DBG = 0, bool WIDE = false, bool PAIR = false, bool GROUP = false>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_pp_kernel(const bf16_t* A_, const bf16_t* B_, long lda_, long ldb_, int M_, int K_, int tiles_m_,
                                                         int tiles_n_, int pp_xr_, int pp_bc_, GemmParams p, std::conditional_t<GROUP, PPGroup, PPNoGroup> grp) {
  p.A = A_; p.B = B_; p.lda = lda_; p.ldb = ldb_; p.M = M_; p.K = K_; p.tiles_m = tiles_m_; p.tiles_n = tiles_n_; p.pp_xr = pp_xr_; p.pp_bc = pp_bc_;
  static_assert(!AT || BT, "the A^T form (weight gradient) has both operands K-strided");
  static_assert(!GROUP || (AT && EPI == EPI_SLAB_F32 && !PAIR), "grouped launches exist for the weight-gradient form only");
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
  unsigned long long rt[5] = {0, 0, 0, 0, 0};                  // (DBG 9: 100-MHz real-time stamps of this workgroup's life)
  if (DBG == 9) rt[0] = __builtin_amdgcn_s_memrealtime();

  // XCD-aware tile map: blocks with equal bid % 8 share an XCD (round-robin dispatch; speed only); each XCD walks a
  // contiguous chunk of the row-major tile order (bijective for any tile count)
  // One tile per workgroup.  (A persistent loop over tiles was built and measured: the next tile's first counted vmcnt
  // wait then absorbs the previous tile's store acknowledgements, and those arrive at the HBM write rate -- an XCD's 32 CUs
  // write more per round than its L2 holds -- so nothing overlapped, and the loop-carried state cost 40-60 VGPRs.)
  int tm, tn, kz = 0;                                            // kz = K slice (weight-gradient form), not blockIdx.z: see below
  int g_kt0 = 0, g_kt1 = 0;                                      // (GROUP: this item's K tiles)
  if constexpr (GROUP) {
    const unsigned code = grp.item[blockIdx.x];
    const int gi = (int)(code & 3u), sl = (int)((code >> 11) & 15u);
    tm = (int)((code >> 2) & 15u); tn = (int)((code >> 6) & 31u);
    // (a select chain, not grp.prob[gi]: a dynamically indexed by-value argument may be copied to scratch memory)
    const PPGroupProb pr = gi == 0 ? grp.prob[0] : gi == 1 ? grp.prob[1] : gi == 2 ? grp.prob[2] : grp.prob[3];
    p.A = pr.A; p.B = pr.B; p.lda = pr.M; p.ldb = pr.N; p.M = pr.M; p.N = pr.N; p.ldc = pr.N; p.outf = pr.out; p.colsum_a = pr.colsum;
    p.K = grp.nk << 6;
    g_kt1 = grp.nk;
    if (sl) {                                                    // one K slice of a split tile: compact partial tile [256][96 NPN] in the workspace
      const long j = (long)blockIdx.x - grp.n_full;
      p.ldc = 96 * NPN;
      p.outf = grp.part + j * (256 * 96 * NPN) - ((long)tm * 256 * (96 * NPN) + (long)tn * (96 * NPN));
      if (p.colsum_a) p.colsum_a = grp.cs_part + j * 256 - (long)tm * 256;
      g_kt0 = ((sl - 1) * grp.nk) / grp.s; g_kt1 = (sl * grp.nk) / grp.s;      // (K tiles dealt to the slices as evenly as possible)
    }
  } else {
    const int tiles = p.tiles_m * p.tiles_n;
    // position in dispatch order (x fastest, then z); workgroups are dealt to the XCDs round-robin in that order
    // (row-major-A forms with an internal K split, gridDim.z > 1: every K slice walks the same tile map; kz = blockIdx.z)
    const int flat = AT ? (int)blockIdx.x + (int)blockIdx.z * tiles : (int)blockIdx.x, nwg = AT ? tiles * (int)gridDim.z : tiles;
    const int xcd = flat & 7, qq = nwg >> 3, rr = nwg & 7;
    const int item = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (flat >> 3);   // contiguous chunk per XCD
    if (!AT) kz = (int)blockIdx.z;
    if (AT) {
      // weight gradient: items ordered (K slice, N tile, M tile) -- an XCD's ~nwg/8 workgroups share ONE K slice of both operands
      // (or two neighbouring ones) and a few tile columns, instead of touching every slice (PMC: 180 MB fetched per launch for 63 MB
      // of operands with the x-then-z order)
      kz = item / tiles;
      const int t = item - kz * tiles;
      tn = t / p.tiles_m; tm = t - tn * p.tiles_m;
    } else if (p.pp_xr) {
      const int xc = 8 / p.pp_xr, R = p.tiles_m / p.pp_xr, C = p.tiles_n / xc, local = flat >> 3;
      const int xi = xcd / xc, xj = xcd - xi * xc;
      const int per = R * p.pp_bc, blk = local / per, rem = local - blk * per;
      const int r = rem / p.pp_bc;
      tm = xi * R + r; tn = xj * C + blk * p.pp_bc + (rem - r * p.pp_bc);
    } else {
      tm = item / p.tiles_n; tn = item - tm * p.tiles_n;
    }
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
  const int kt0 = GROUP ? g_kt0 : (int)(((long)kz * nk_all) / gridDim.z), kt1 = GROUP ? g_kt1 : (int)(((long)(kz + 1) * nk_all) / gridDim.z);
  const int nk = kt1 - kt0;
  const long a_step = AT ? 64 * p.lda * 2 : 128, b_step = BT ? 64 * p.ldb * 2 : 128;
  const char* a_ptr = (const char*)(AT ? p.A + m0 : p.A + m0 * p.lda) + kt0 * a_step;      // first K tile of the slice
  const char* b_ptr = (const char*)(BT ? p.B + n0 : p.B + n0 * p.ldb) + kt0 * b_step;
  const long b_part_step = BT ? 48 * 2 : 48 * p.ldb * 2;       // part j -> j + 1

  // issue unit `u` (0/1 = A halves, 2+j = B parts) of K tile (t + d) into LDS stage `stg`; ap / bp = pointers of tile t
  auto issue = [&](auto U, const char* ap, const char* bp, int d, int stg) {
    constexpr int u = decltype(U)::value;
    char* sb = smem + stg * STAGE;
#ifdef CAREL_PP_HOT_TILE     // timing ablation (tagged build only, results wrong): every K tile re-reads the slice's FIRST tile -- L2-hot after the first touch
    ap = a_ptr; bp = b_ptr; d = 0;
#endif
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

  // GELU epilogues: the table goes into the LDS behind the staging buffers by 20 LDS-DMA pieces issued BEFORE every operand copy (vmcnt
  // retires in order: the counted waits of the schedule are unchanged, and the first of them already covers the table)
  constexpr bool LUT = epi_is_gelu(EPI) && !AT;
  const uint32_t* lut_lds = (LUT && p.gelu_lut) ? (const uint32_t*)(smem + G::LDS) : nullptr;
  if (LUT && p.gelu_lut) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int piece = wave * 3 + k;                          // wave-uniform
      if (piece * 1024 < LUT_BYTES) {
        __builtin_amdgcn_global_load_lds((const void*)((const char*)g_gelu_lut + piece * 1024 + lane * 16), (CAREL_LDS void*)(smem + G::LDS + piece * 1024), 16, 0, 0);
        asm volatile("" ::: "memory");
      }
    }
  }
  // ---- epilogue geometry (needed early: the epilogue's INPUTS are requested before the main loop) -----------------------
  const int rho = lane >> 4;
  constexpr int NQ = NF / 2;
  auto row_of = [&](int b) { return m0 + (AT ? (b >> 1) * 128 + wr * 32 : wr * 64 + (b >> 1) * 32) + (b & 1) * 16 + (lane & 15); };
  // Epilogue inputs ahead of the main loop.  The residual rows / saved gelu'(u) values an epilogue combines with the accumulators do
  // not depend on them, and in a training step they are COLD (written a forward pass ago, or by a kernel ~100 MB of traffic ago):
  // requested after the last K tile they cost one exposed HBM round trip per row block, and the odd third fragment of the 96-column
  // tile used to load -> wait -> store four times in a row (tools/bench_gemm_cold.py: +3.0 .. +5.4 us per launch for cold inputs).
  // So the first PB row blocks' inputs are requested right behind the prologue's DMA units and sit in registers through the main loop
  // (NPN 1: all four blocks, 48 registers; NPN 2: all four for the bf16 aux values of the FFN2 data gradient, 48 registers, two for f32 residuals).  vmcnt retires in issue order, so every counted wait that targets a
  // PROLOGUE unit grows by the NE load instructions issued after it: the prologue's own wait and the waits of K tile 0 (all of which
  // target prologue units -- checked by tools/gemm_sched.py, first_tile_waits_target_prologue); from tile 1 on the waits target units
  // issued after these loads and are unchanged, i.e. the inputs get one K tile of flight before anything waits for them.  Rows past M
  // are clamped (never stored), so the instruction count is static.
  constexpr bool HAS_IN = EPI == EPI_BIAS_DROP_RESID || EPI == EPI_ADD_F32 || epi_is_dgelu(EPI);
  constexpr int PB = (HAS_IN && !AT && WIDE && DBG == 0) ? (NPN == 1 ? 4 : NPN == 2 ? (epi_is_dgelu(EPI) ? 4 : 2) : 0) : 0;    // (bf16 aux: 4 registers per 8 columns)
  constexpr int NE = PB * (NQ * (epi_is_dgelu(EPI) ? 1 : 2) + (NF & 1));
  static_assert(NE == 0 || (S::NTAIL >= 1 && NE + 16 <= 63), "vmcnt is a 6-bit counter");
  const bool pre = PB > 0 && p.epi_prefetch && (EPI != EPI_ADD_F32 || p.resid != nullptr) && nk > S::NTAIL && !(PAIR && kz == 0);      // wave-uniform
  EpiIn8 pin[PB > 0 ? PB : 1][NQ > 0 ? NQ : 1];
  EpiIn4 pin4[PB > 0 ? PB : 1];
  // ---- prologue: the units the steady-state schedule would have issued before phase 0 ------------------------------
  static_for<S::NPRO>([&](auto I) {
    constexpr int i = decltype(I)::value;
    issue(IC<S::pro_unit[i]>{}, a_ptr, b_ptr, S::pro_tile[i], S::pro_tile[i] % ST);
  });
  if constexpr (PB > 0) {
    if (pre) {
      static_for<PB>([&](auto BB) {
        constexpr int b = decltype(BB)::value;
        long row = row_of(b);
        if (row > (long)p.M - 1) row = (long)p.M - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) epi_in8<EPI, NPN == 1>(p, row, n0 + wc * WN + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8, pin[b][q]);
        if constexpr (NF & 1) epi_in4<EPI, NPN == 1>(p, row, n0 + wc * WN + (NF - 1) * 16 + rho * 4, pin4[b]);
      });
      asm volatile("" ::: "memory");
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::PRO_WAIT + NE) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::PRO_WAIT) : "memory");
    }
  } else {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::PRO_WAIT) : "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wc == 1) __builtin_amdgcn_s_barrier();                   // group 1 runs one barrier behind group 0

  int sidx = 0;                                                // LDS stage of the current K tile
  bool first_tile = true;                                      // (ablation builds only)
  // one K tile; R = 0: steady state, R = r > 0: r tiles remain including this one (tail vmcnt tables, no issue past K)
  int tile_no = 0;                                             // (DBG 9: in-kernel stamps of one steady-state K tile, workgroup 0)
  auto tile = [&](auto RR, auto XX) {
    constexpr int R = decltype(RR)::value, XW = decltype(XX)::value;      // XW: load instructions in flight that are NEWER than every unit this tile waits for
    const char* st = smem + sidx * STAGE;
    const bool stamp = DBG == 9 && blockIdx.x == 0 && tile_no == (nk >> 1) && (wave == 0 || wave == 4);
    static_for<NP>([&](auto PP) {
      constexpr int P = decltype(PP)::value;
      constexpr int h = S::phase_h[P], j = S::phase_j[P];
      unsigned long long ts[8];
      if (DBG == 9) ts[0] = __builtin_amdgcn_s_memtime();
      // ---------------- load segment L(P): fragments of this phase, this phase's DMA units, counted wait -------------
      const bool do_reads = (DBG != 3 && DBG != 6 && DBG != 7 && DBG != 8) || first_tile;     // 6: DMA + barriers only, 7: barriers only, 8: MFMA + barriers only
      // the phase's DMA unit e (tile t + delta); the wide NT schedule interleaves the units with the fragment-read groups: the four waves of a group
      // saturate the LDS for ~270 cycles with their reads (in-kernel stamps, tools/stamp_gemm_pp.py), and an LDS-DMA instruction
      // issued in between costs the wave its ~30 cycles of issue INSIDE that time instead of after it
      // (row-image operands only, i.e. the NT form: with K-strided operands -- ds_read_b64_tr_b16 fragment reads -- the same interleave
      // made the NN / TN forms 20-40 % SLOWER: measured, tools/bench_gemm_pp.py)
      constexpr bool IL = WIDE && !AT && !BT;
      auto issue_e = [&](auto E) {
        constexpr int e = decltype(E)::value;
        if constexpr (e < S::n_issue[P]) {
          constexpr int u = S::issue_unit[P][e], d = S::issue_delta[P][e];
          if constexpr ((R == 0 || d < R) && DBG != 1 && DBG != 7 && DBG != 8) {
            int stg = sidx + d;
            if (stg >= ST) stg -= ST;
            issue(IC<u>{}, a_ptr, b_ptr, d, stg);
          }
        }
      };
      if constexpr (WIDE) {
        if constexpr (P == 0) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            if (do_reads)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int ks = 0; ks < 2; ++ks)
                fa[hh][i][ks] = AT ? frag16_col(st + hh * 16384, wr * 32 + i * 16, ks * 32) : frag16_row(st, wr * 64 + hh * 32 + i * 16, ks * 32);
            if constexpr (IL) { if (hh == 0) issue_e(IC<0>{}); else issue_e(IC<1>{}); }
          }
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
      if (DBG == 9) { __builtin_amdgcn_sched_barrier(0); ts[1] = __builtin_amdgcn_s_memtime(); }       // fragment reads issued
      // the units not placed between read groups above (all of them for the fine schedule and for WIDE phases > 0)
      static_for<S::MAXI>([&](auto E) {
        constexpr int e = decltype(E)::value;
        if constexpr (!(IL && P == 0 && e < 2)) issue_e(IC<e>{});
      });
      if (DBG == 9) { __builtin_amdgcn_sched_barrier(0); ts[2] = __builtin_amdgcn_s_memtime(); }       // DMA issued
      if constexpr (S::wait[R][P] >= 0 && DBG != 1 && DBG != 7 && DBG != 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::wait[R][P] + XW) : "memory");
      if (DBG == 9) { __builtin_amdgcn_sched_barrier(0); ts[3] = __builtin_amdgcn_s_memtime(); }       // counted vmcnt wait over
      if constexpr (WIDE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads complete before the barrier: war = 1
      if (DBG == 9) { __builtin_amdgcn_sched_barrier(0); ts[4] = __builtin_amdgcn_s_memtime(); }       // fragments landed
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (DBG == 9) ts[5] = __builtin_amdgcn_s_memtime();                                              // barrier passed: matrix segment starts
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
      if (DBG == 9) ts[6] = __builtin_amdgcn_s_memtime();                                              // MFMAs issued
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (DBG == 9) {
        ts[7] = __builtin_amdgcn_s_memtime();
        if (stamp && lane == 0 && p.splitk_ws) {
          unsigned long long* o = (unsigned long long*)p.splitk_ws + ((wave >> 2) * 6 + P) * 8;
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = ts[i];
        }
      }
    });
    a_ptr += a_step; b_ptr += b_step;
    sidx = sidx + 1 == ST ? 0 : sidx + 1;
    first_tile = false;
    ++tile_no;
  };
  unsigned long long t_loop0 = 0;
  if (DBG == 9) { t_loop0 = __builtin_amdgcn_s_memtime(); rt[1] = __builtin_amdgcn_s_memrealtime(); }
  int t_first = 0;
  if constexpr (NE > 0) {
    if (pre) { tile(IC<0>{}, IC<NE>{}); t_first = 1; }         // K tile 0: its waits target prologue units, older than the NE input loads
  }
  for (int t = t_first; t < nk - S::NTAIL; ++t) tile(IC<0>{}, IC<0>{});
  static_for<S::NTAIL>([&](auto I) { tile(IC<S::NTAIL - decltype(I)::value>{}, IC<0>{}); });
  if (wc == 0) __builtin_amdgcn_s_barrier();                   // both groups have now passed the same number of barriers
  if (DBG == 9) rt[2] = __builtin_amdgcn_s_memrealtime();
  if (DBG == 9 && p.splitk_ws && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
    unsigned long long* o = (unsigned long long*)p.splitk_ws + 200 + (blockIdx.x ? 4 : 0);
    o[0] = t_loop0; o[1] = __builtin_amdgcn_s_memtime(); o[2] = nk;
  }

  if constexpr (PAIR) {
    constexpr int NR = 4 * NF;                                  // float4 accumulators per lane
    const long slot = ((long)(tm * p.tiles_n + tn) * 8 + wave);
    unsigned long long* part = (unsigned long long*)p.splitk_ws + slot * (NR * 2 * 64) + lane;      // [register half][lane] 8-byte words
    unsigned* flag = p.pair_flags + slot;
    if (kz == 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NF; ++j) {
            const int r = (h * 2 + i) * NF + j;
            const f32x4 a = acc[h][i][j];
            __hip_atomic_store(part + (2 * r) * 64, ((unsigned long long)__float_as_uint(a[1]) << 32) | __float_as_uint(a[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(part + (2 * r + 1) * 64, ((unsigned long long)__float_as_uint(a[3]) << 32) | __float_as_uint(a[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every partial sum of this wave is written through
      __hip_atomic_store(flag, p.pair_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    // slice 1: wait for THIS wave's partner (bounded: ~2 s of polling, then the result is poisoned instead of hanging the GPU)
    unsigned seen = 0;
    for (int spin = 0; spin < (1 << 22); ++spin) {
      seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (seen == p.pair_seq) break;
      __builtin_amdgcn_s_sleep(16);
    }
    asm volatile("" ::: "memory");
    const float poison = seen == p.pair_seq ? 0.f : __builtin_nanf("");
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        unsigned long long w[NF][2];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const int r = (h * 2 + i) * NF + j;
          w[j][0] = __hip_atomic_load(part + (2 * r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          w[j][1] = __hip_atomic_load(part + (2 * r + 1) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          // (first half of K) + (second half of K): the partner's sum is the left operand
          acc[h][i][j][0] = (__uint_as_float((unsigned)w[j][0]) + acc[h][i][j][0]) + poison;
          acc[h][i][j][1] = (__uint_as_float((unsigned)(w[j][0] >> 32)) + acc[h][i][j][1]) + poison;
          acc[h][i][j][2] = (__uint_as_float((unsigned)w[j][1]) + acc[h][i][j][2]) + poison;
          acc[h][i][j][3] = (__uint_as_float((unsigned)(w[j][1] >> 32)) + acc[h][i][j][3]) + poison;
        }
      }
  }
  if (AT && do_cs && lane < 16) {                              // D[n][m]: every row n holds the same sum; lane = m
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) p.colsum_a[(long)kz * p.M + m0 + h * 128 + wr * 32 + i * 16 + lane] = acc1[h][i][0];
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
  if (EPI == EPI_SLAB_F32) p.outf += ((long)kz - (long)blockIdx.z) * p.M * p.ldc;      // the shared epilogue indexes slabs by blockIdx.z
  float cs[NQ > 0 ? NQ : 1][8];
#pragma unroll
  for (int q = 0; q < (NQ > 0 ? NQ : 1); ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[q][e] = 0.f;
  float bias8[NQ > 0 ? NQ : 1][8];
#pragma unroll
  for (int q = 0; q < NQ; ++q) epi_bias8<EPI>(p, n0 + wc * WN + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8, bias8[q]);
  // (EPI_BIAS_DROP_RESID with a recomputed LayerNorm residual: gamma / beta of this lane's columns, once, up front -- on the 96-wide tile only
  // (the encoder's shape: 24 registers); the wider tiles have no registers for them and never get such a GEMM: gemm_pp_pick)
  constexpr bool LN_HOIST = NPN == 1;
  float ln8[LN_HOIST && NQ > 0 ? NQ : 1][16], ln4[8];
  if constexpr (LN_HOIST) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) epi_ln8<EPI>(p, n0 + wc * WN + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8, ln8[q]);
  }
  if constexpr ((NF & 1) && LN_HOIST) epi_ln4<EPI>(p, n0 + wc * WN + (NF - 1) * 16 + rho * 4, ln4);
  // (npn 3 with a residual / aux input has no registers for two blocks of inputs: load and use block by block there;
  // the encoder never runs that combination -- N = 2304 is the bias-only QKV projection)
  constexpr bool PIPE = NPN < 3 || EPI == EPI_BIAS_BF16 || epi_is_gelu(EPI) || EPI == EPI_SLAB_F32;
  EpiIn8 in[PIPE ? 2 : 1][NQ > 0 ? NQ : 1];
  auto load_block = [&](int b, EpiIn8* dst) {
    const long row = row_of(b);
    if (row < (long)p.M) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) epi_in8<EPI, NPN == 1>(p, row, n0 + wc * WN + (2 * q + (rho & 1)) * 16 + (rho >> 1) * 8, dst[q]);
    }
  };
  // blocks below PB already hold their inputs (requested before the main loop) when `pre`
  if (PIPE && !(pre && PB > 0)) load_block(0, in[0]);
  static_for<4>([&](auto BB) {
    constexpr int b = decltype(BB)::value;
    if constexpr (PIPE) { if (b + 1 < 4 && !(pre && b + 1 < PB)) load_block(b + 1, in[(b + 1) & 1]); }
    else { if (!(pre && b < PB)) load_block(b, in[0]); }
    constexpr int h = b >> 1, i = b & 1;
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
      const float* lnp = ln8[LN_HOIST ? q : 0];
      if (ok) {
        bool done = false;         // (no struct copy / select here: a copied EpiIn8 ends up in scratch memory)
        if constexpr (b < PB) { if (pre) { epi_out8<EPI, NPN == 1>(p, v, bias8[q], pin[b][q], row, col, lut_lds, lnp); done = true; } }
        if (!done) epi_out8<EPI, NPN == 1>(p, v, bias8[q], in[PIPE ? (b & 1) : 0][q], row, col, lut_lds, lnp);
        if (epi_is_dgelu(EPI)) {
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[q][e] += v[e];
        }
      }
    }
  });
  if constexpr (NF & 1) {      // the odd last fragment: 4 columns per lane, after all paired stores
    static_for<4>([&](auto BB) {
      constexpr int b = decltype(BB)::value;
      const long row = row_of(b);
      if (row < (long)p.M) {
        const long col = n0 + wc * WN + (NF - 1) * 16 + rho * 4;
        bool done = false;
        if constexpr (b < PB) { if (pre) { epi_out4<EPI, NPN == 1>(p, acc[b >> 1][b & 1][NF - 1], pin4[b], row, col, ln4); done = true; } }
        if (!done) epi_store<EPI, NPN == 1>(p, acc[b >> 1][b & 1][NF - 1], row, col);
      }
    });
  }
  if (DBG == 9) {
    rt[3] = __builtin_amdgcn_s_memrealtime();                  // every store of this wave issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rt[4] = __builtin_amdgcn_s_memrealtime();                  // ... and acknowledged
    if (p.splitk_ws && lane == 0 && wave == 0) {
      unsigned long long* o = (unsigned long long*)p.splitk_ws + 512 + (long)blockIdx.x * 5;
#pragma unroll
      for (int i = 0; i < 5; ++i) o[i] = rt[i];
    }
  }
  if (epi_is_dgelu(EPI) && p.colsum_part) {                // block-uniform; dispatcher guarantees NF even here
    // per-128-row column sums of the stored values (the FFN1 bias gradient): 16 lanes -> 1, then wave rows 2r, 2r+1
    float* sc = (float*)smem;                                  // [4 wave rows][BN]; every LDS read / DMA has retired
#pragma unroll
    for (int q = 0; q < NF / 2; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = row16_sum(cs[q][e]);         // (24 values x 4 ds_bpermute cost 3.7 us per tile: tools/bench_epilogues.py)
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

template <int NPN, bool AT, bool BT, int EPI, int DBG = 0, bool WIDE = false, bool PAIR = false>
int launch_pp(GemmParams p, int splits, hipStream_t s) {
  using G = PPGeom<NPN, BT>;
  constexpr int LDS_BYTES = G::LDS + ((epi_is_gelu(EPI) && !AT) ? LUT_BYTES : 0);
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  if (epi_is_gelu(EPI)) { const int rc = gelu_lut_ready(); if (rc) return rc; }
  static bool attr = false;      // per process; setting it again is harmless if two threads race
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_pp_kernel<NPN, AT, BT, EPI, DBG, WIDE, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "gemm_pp_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr = true;
  }
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = p.N / (96 * NPN);
  p.gelu_lut = g_pp_gelu_lut;
  p.epi_prefetch = g_pp_epi_prefetch;
  p.pp_xr = 0; p.pp_bc = 1;
  if (!AT && g_pp_xcd_rect) {
    // XCD rectangles: the partition xr x (8 / xr) of the tile grid whose rectangles stage the fewest operand rows (R x 256 of A plus
    // C x 96 npn of B), each walked in column blocks so that one round (32 tiles per XCD) is compact too: FFN1 forward then keeps
    // half of W1 (2.4 MB) + four row blocks of A (1.5 MB) in a 4-MiB L2 per round instead of streaming all of W1 through it twice
    long best = -1;
    for (int xr = 8; xr >= 1; xr >>= 1) {
      const int xc = 8 / xr;
      if (p.tiles_m % xr || p.tiles_n % xc) continue;
      const long cost = (long)(p.tiles_m / xr) * 256 + (long)(p.tiles_n / xc) * 96 * NPN;
      if (best < 0 || cost < best) { best = cost; p.pp_xr = xr; }
    }
    if (p.pp_xr) {
      const int R = p.tiles_m / p.pp_xr, C = p.tiles_n / (8 / p.pp_xr);
      int bc = 1;
      for (int d = 1; d <= C; ++d) if (C % d == 0 && (long)R * d <= 32) bc = d;
      p.pp_bc = bc;
    }
  }
  hipLaunchKernelGGL((gemm_pp_kernel<NPN, AT, BT, EPI, DBG, WIDE, PAIR>), dim3(p.tiles_m * p.tiles_n, 1, splits), dim3(512), LDS_BYTES, s, p.A, p.B, p.lda,
                     p.ldb, p.M, p.K, p.tiles_m, p.tiles_n, p.pp_xr, p.pp_bc, p, PPNoGroup{});
  return check_launch("gemm_pp_kernel");
}

// ---- grouped weight gradients (GROUP): reduction of the split tiles + the small sums that ride along ---------------------------------
// Blocks [0, P * TB): split tile t = b / TB, 256 float4 of it per block: out = sum over the s slices IN SLICE ORDER (fixed order of additions:
// bit-reproducible) of the compact partial tiles, written to the tile's place in dW.  Blocks [P * TB, P * TB + P): the bias-gradient
// partials of split tile t (first tile column of a problem with a bias gradient only).  The blocks after that sum the per-block partials of up
// to two LayerNorm backward passes into dgamma / dbeta / bias gradient (partial_colsum16: the device function, and the bits, of
// partial_reduce_seg_kernel) -- they used to ride on the per-GEMM slab reductions.
struct PPGroupLn { const float* partials; SegOuts outs; int nparts; };
template <int NPN>
__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(PPGroup grp, PPGroupLn ln0, PPGroupLn ln1) {
  constexpr int BN = 96 * NPN, TILE = 256 * BN, TB = TILE / 4 / 256;
  __shared__ float lds[256];
  const int P = grp.s > 0 ? (grp.n_items - grp.n_full) / grp.s : 0;
  int b = (int)blockIdx.x;
  if (b < P * TB + P) {
    const bool cs = b >= P * TB;
    const int t = cs ? b - P * TB : b / TB;
    const unsigned code = grp.item[grp.n_full + t * grp.s];
    const int gi = (int)(code & 3u), tm = (int)((code >> 2) & 15u), tn = (int)((code >> 6) & 31u);
    const PPGroupProb pr = gi == 0 ? grp.prob[0] : gi == 1 ? grp.prob[1] : gi == 2 ? grp.prob[2] : grp.prob[3];
    if (cs) {
      if (!pr.colsum || tn != 0) return;
      const float* src = grp.cs_part + (long)t * grp.s * 256 + threadIdx.x;
      float a = src[0];
      for (int z = 1; z < grp.s; ++z) a += src[(long)z * 256];
      pr.colsum[tm * 256 + threadIdx.x] = a;
      return;
    }
    const int i4 = (b - t * TB) * 256 + (int)threadIdx.x;          // float4 index inside the tile
    const int r = i4 / (BN / 4), c4 = i4 - r * (BN / 4);
    const float* src = grp.part + (long)t * grp.s * TILE + (long)i4 * 4;
    float4 a = *(const float4*)src;
    int z = 1;
    for (; z + 3 < grp.s; z += 4) {                                  // four slices requested before the first add (same order of additions)
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(src + (long)(z + u) * TILE);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; z < grp.s; ++z) {
      const float4 v = *(const float4*)(src + (long)z * TILE);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *(float4*)(pr.out + ((long)tm * 256 + r) * pr.N + (long)tn * BN + c4 * 4) = a;
    return;
  }
  b -= P * TB + P;
  constexpr int LNB = (3 * 768 + PR_COLS - 1) / PR_COLS;
  const PPGroupLn& ln = b < LNB ? ln0 : ln1;
  if (b >= LNB) b -= LNB;
  if (!ln.partials) return;
  int c;
  const float tsum = partial_colsum16(ln.partials, 3 * 768, ln.nparts, lds, c, b);
  if (threadIdx.x < PR_COLS && c < 3 * 768) {
    float* o = ln.outs.p[c / 768];
    if (o) o[c % 768] = tsum;
  }
}

// the work list of a grouped launch (see PPGroup): `cus` whole tiles per round first, each XCD's share a run of consecutive tiles in
// (problem, tile row, tile column) order; the remaining tiles split into s K slices each
static int wgrad_group_plan(const WgradGroupProb* pb, int n, long T, PPGroup& g, int npn, bool allow_split = true) {
  const int BN = 96 * npn, cus = 256;
  if (n < 1 || n > 4 || T < 256 || (T & 63)) return 0;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (pb[i].M < 256 || pb[i].M % 256 || pb[i].N < BN || pb[i].N % BN || pb[i].M / 256 > 16 || pb[i].N / BN > 32) return 0;
    tiles += (pb[i].M / 256) * (pb[i].N / BN);
  }
  const int nk = (int)(T >> 6);
  int n_full = (tiles / cus) * cus, P = tiles - n_full, s = 0;
  if (P > 0 && !allow_split) { n_full = tiles; P = 0; }
  if (P > 0) {
    s = cus / P;                                   // the split items fill one more round of the chip ...
    if (s > 8) s = 8;
    if (s > nk / 4) s = nk / 4;                    // ... with at least four K tiles each (the static schedule's prologue + tail)
    if (s < 2) { s = 0; n_full = tiles; P = 0; }   // not worth splitting: the remainder runs as whole tiles in a last partial round
  }
  if (nk < 4 || n_full + P * s > PP_GROUP_MAX_ITEMS) return 0;
  unsigned short lin[PP_GROUP_MAX_ITEMS];
  int k = 0;
  // linear tile order per problem: an XCD's run of 32 consecutive tiles should stage few distinct operand blocks (256 dY columns per tile row,
  // 96 X columns per tile column).  Few tile rows (FFN2: 3 x 32) -> tile rows innermost: a run is all 3 rows x ~11 columns (233 KB per K
  // tile into that XCD's L2 instead of 425 KB for 1 row x 32 columns); otherwise tile columns innermost (FFN1 12 x 8, QKV 9 x 8: 4 rows x 8 columns)
  for (int i = 0; i < n; ++i) {
    const int ntm = pb[i].M / 256, ntn = pb[i].N / BN;
    if (ntm <= 4 && ntn > 8) {
      for (int tn = 0; tn < ntn; ++tn)
        for (int tm = 0; tm < ntm; ++tm) lin[k++] = (unsigned short)(i | (tm << 2) | (tn << 6));
    } else {
      for (int tm = 0; tm < ntm; ++tm)
        for (int tn = 0; tn < ntn; ++tn) lin[k++] = (unsigned short)(i | (tm << 2) | (tn << 6));
    }
  }
  // whole tiles: round r (cus tiles) of the linear order, XCD x takes the run [x * cus / 8, (x + 1) * cus / 8) of it; dispatch slot 8 q + x
  const int per = cus / 8, full_rounds = (n_full / cus) * cus;
  for (int f = 0; f < full_rounds; ++f) {
    const int r = f / cus, w = f - r * cus, x = w & 7, q = w >> 3;
    g.item[f] = lin[r * cus + x * per + q];
  }
  for (int f = full_rounds; f < n_full; ++f) g.item[f] = lin[f];      // a last partial round of whole tiles (s = 0 above): plain order
  for (int t = 0; t < P; ++t)
    for (int z = 0; z < s; ++z) g.item[n_full + t * s + z] = (unsigned short)(lin[n_full + t] | ((z + 1) << 11));
  for (int i = 0; i < 4; ++i) {
    const WgradGroupProb& q = pb[i < n ? i : 0];
    g.prob[i] = PPGroupProb{(const bf16_t*)q.dY, (const bf16_t*)q.X, (float*)q.dW, (float*)q.db, q.M, q.N};
  }
  g.nk = nk; g.s = s; g.n_full = n_full; g.n_items = n_full + P * s;
  return 1;
}

CAREL_TUNABLE(int, g_pp_wide, 1);        // tuning hook (carel_gemm_set_variant(90 / 91)): the fine (12-MFMA phases) / wide-phase schedule

template <bool BT, int EPI>
int launch_pp_n(const GemmParams& p, int npn, hipStream_t s) {
  if (npn == 1) return g_pp_wide ? launch_pp<1, false, BT, EPI, 0, true>(p, 1, s) : launch_pp<1, false, BT, EPI>(p, 1, s);
  if (npn == 2) return g_pp_wide ? launch_pp<2, false, BT, EPI, 0, true>(p, 1, s) : launch_pp<2, false, BT, EPI>(p, 1, s);
  if constexpr (!BT) { if (npn == 3) return g_pp_wide ? launch_pp<3, false, BT, EPI, 0, true>(p, 1, s) : launch_pp<3, false, BT, EPI>(p, 1, s); }
  return set_error(CAREL_ERR_ARG, "gemm_pp_launch: npn = %d not built for this form", npn);
}

}  // namespace

size_t gemm_pp_wgrad_group_ws_bytes(const WgradGroupProb* pb, int n, long T) {
  PPGroup g;
  if (!wgrad_group_plan(pb, n, T, g, 1)) return 0;
  return (size_t)(g.n_items - g.n_full) * (256 * 96 + 256) * 4 + 256;
}

// 0 = this problem list cannot run grouped (the caller falls back to one GEMM per weight gradient)
int gemm_pp_wgrad_group_ok(const WgradGroupProb* pb, int n, long T) {
  PPGroup g;
  return wgrad_group_plan(pb, n, T, g, 1);
}

CAREL_TUNABLE(int, g_group_mode, 0);       // (experiments, hook 250 + m) 0 = 256 x 96 tiles, remainder split along K (product); 1 = 256 x 192 whole tiles; 2 = 256 x 96 whole tiles
#ifdef CAREL_EXPERIMENTS
void gemm_pp_group_mode(int m) { g_group_mode = (m >= 0 && m <= 2) ? m : 0; }
#endif
template <int NPN>
static int wgrad_group_launch(const WgradGroupProb* pb, int n, long T, void* ws, size_t ws_bytes, const WgradGroupLn* ln, int n_ln, bool allow_split, hipStream_t stream) {
  PPGroup g;
  if (!pb || !wgrad_group_plan(pb, n, T, g, NPN, allow_split)) return set_error(CAREL_ERR_SHAPE, "carel_gemm_wgrad_group: 1-4 problems with M a multiple of 256 (<= 4096), N of 96 (<= 3072), T of 64 (>= 256)");
  for (int i = 0; i < n; ++i) {
    if (!pb[i].dY || !pb[i].X || !pb[i].dW) return set_error(CAREL_ERR_ARG, "carel_gemm_wgrad_group: null operand");
    if (((uintptr_t)pb[i].dY | (uintptr_t)pb[i].X | (uintptr_t)pb[i].dW) & 15) return set_error(CAREL_ERR_ARG, "carel_gemm_wgrad_group: operands must be 16-byte aligned");
  }
  const int n_split = g.n_items - g.n_full;
  const size_t need = (size_t)n_split * (256 * 96 * NPN + 256) * 4 + 256;
  if (n_split && (!ws || ws_bytes < need || ((uintptr_t)ws & 15))) return set_error(CAREL_ERR_ARG, "carel_gemm_wgrad_group: workspace of %zu bytes needed (carel_gemm_wgrad_group_ws_bytes)", need);
  if (n_ln < 0 || n_ln > 2 || (n_ln && !ln)) return set_error(CAREL_ERR_ARG, "carel_gemm_wgrad_group: at most two LayerNorm partial sets");
  g.part = (float*)ws;
  g.cs_part = g.part + (size_t)n_split * 256 * 96 * NPN;
  using G = PPGeom<NPN, true>;
  constexpr int LDS_BYTES = G::LDS;
  auto kern = gemm_pp_kernel<NPN, true, true, EPI_SLAB_F32, 0, true, false, true>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "gemm_pp_kernel (grouped): hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr = true;
  }
  GemmParams p = {};
  p.gelu_lut = 0; p.epi_prefetch = 0; p.pp_bc = 1;
  hipLaunchKernelGGL(kern, dim3(g.n_items), dim3(512), LDS_BYTES, stream, (const bf16_t*)nullptr, (const bf16_t*)nullptr, 0L, 0L, 0, 0, 0, 0, 0, 1, p, g);
  int rc = check_launch("gemm_pp_kernel (grouped weight gradients)");
  if (rc) return rc;
  PPGroupLn l[2] = {};
  for (int i = 0; i < n_ln; ++i) {
    l[i].partials = (const float*)ln[i].partials; l[i].nparts = ln[i].nparts;
    l[i].outs.p[0] = (float*)ln[i].dgamma; l[i].outs.p[1] = (float*)ln[i].dbeta; l[i].outs.p[2] = (float*)ln[i].dbias; l[i].outs.p[3] = nullptr;
  }
  constexpr int TB = 256 * 96 * NPN / 4 / 256, LNB = (3 * 768 + PR_COLS - 1) / PR_COLS;
  const int P = g.s ? n_split / g.s : 0;
  const int blocks = P * TB + P + n_ln * LNB;
  if (blocks == 0) return CAREL_OK;
  hipLaunchKernelGGL((wgrad_group_reduce_kernel<NPN>), dim3(blocks), dim3(256), 0, stream, g, l[0], l[1]);
  return check_launch("wgrad_group_reduce_kernel");
}
int gemm_pp_wgrad_group(const WgradGroupProb* pb, int n, long T, void* ws, size_t ws_bytes, const WgradGroupLn* ln, int n_ln, hipStream_t stream) {
#ifdef CAREL_EXPERIMENTS
  if (g_group_mode == 1) return wgrad_group_launch<2>(pb, n, T, ws, ws_bytes, ln, n_ln, false, stream);
  if (g_group_mode == 2) return wgrad_group_launch<1>(pb, n, T, ws, ws_bytes, ln, n_ln, false, stream);
#endif
  return wgrad_group_launch<1>(pb, n, T, ws, ws_bytes, ln, n_ln, true, stream);
}

int gemm_pp_init_device(int device) {
  if (device < 0 || device >= 16) return set_error(CAREL_ERR_ARG, "carel_init: device index %d out of range", device);
  if (g_lut_state[device].load(std::memory_order_acquire) == 1) return CAREL_OK;
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_init: hipGetDevice failed");
  if (cur != device && hipSetDevice(device) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_init: hipSetDevice(%d) failed", device);
  hipLaunchKernelGGL(gelu_lut_fill_kernel, dim3((LUT_WORDS + 255) / 256), dim3(256), 0, 0);
  const hipError_t e = hipDeviceSynchronize();
  if (cur != device) (void)hipSetDevice(cur);
  if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_init: GELU table fill failed: %s", hipGetErrorString(e));
  g_lut_state[device].store(1, std::memory_order_release);     // two racing initialisers both fill the same values: harmless
  return CAREL_OK;
}
CAREL_TUNABLE(int, g_pp_force_npn, 0);     // tuning hook (carel_gemm_set_variant(70 + n)): tile width 96 n wherever N allows; 0 = heuristic
#ifdef CAREL_EXPERIMENTS
void gemm_pp_wide_variant(int on) { g_pp_wide = on ? 1 : 0; }
void gemm_pp_xcd_rect(int on) { g_pp_xcd_rect = on ? 1 : 0; }
void gemm_pp_gelu_lut(int on) { g_pp_gelu_lut = on ? 1 : 0; }
void gemm_pp_epi_prefetch(int on) { g_pp_epi_prefetch = on ? 1 : 0; }
void gemm_pp_force_npn(int n) { g_pp_force_npn = (n >= 1 && n <= 3) ? n : 0; }
#endif

// npn (1..3) when the ping-pong kernel should run this GEMM, 0 when it cannot or should not.
int gemm_pp_pick(const GemmParams& p, bool bt, int epi, int force) {
  if (p.N % 96 || p.K % 64 || p.K < 256 || p.M < 1) return 0;
  if (bt ? !(epi == EPI_BIAS_BF16 || epi_is_dgelu(epi) || epi == EPI_ADD_F32)
         : !(epi == EPI_BIAS_BF16 || epi_is_gelu(epi) || epi == EPI_BIAS_DROP_RESID || epi == EPI_ADD_F32)) return 0;
  const int tiles_m = (p.M + 255) / 256;
  int best = 0; double best_score = 0.0;
  for (int npn = bt ? 2 : 3; npn >= 1; --npn) {
    if (p.N % (96 * npn)) continue;
    if (p.resid_stats && npn > 1) continue;                                 // the recomputed LayerNorm residual exists on the 96-wide tile only
    if (g_pp_force_npn && npn != g_pp_force_npn && p.N % (96 * g_pp_force_npn) == 0 && !(bt && g_pp_force_npn == 3)) continue;
    if (epi_is_dgelu(epi) && p.colsum_part && (npn & 1)) continue;      // the fused column sums need fragment pairs
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
    // counted in 256 x 96 tiles over ALL the rows of the split_tile_factor equal GEMMs the caller runs side by side (the forward pass's half-batch
    // chains): whether a GEMM runs here in one pass or split along K on the other path must not depend on how its rows are dealt to chains
    // (a sample's bits would: tools/stress_streams.py), nor on which npn the fill score prefers for this call's own rows
    const int factor = (p.split_tile_factor & 0xff) > 0 ? (p.split_tile_factor & 0xff) : 1;
    const long tiles = (((long)p.M * factor + 255) / 256) * (p.N / 96);
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
CAREL_TUNABLE(int, g_pp_wgrad_force, 0);   // tuning hook (carel_gemm_set_variant(100 + s)): this split-K factor for every supported weight gradient
#ifdef CAREL_EXPERIMENTS
void gemm_pp_wgrad_force(int s) { g_pp_wgrad_force = (s >= 0 && s <= 16) ? s : 0; }
#endif

int gemm_pp_wgrad_splits(int M, int N, long K) {
  if (M % 256 || N % 96 || K % 64 || K < 512) return 0;
  if (g_pp_wgrad_force) { const long m = (K >> 6) / 4; return (int)(g_pp_wgrad_force < m ? g_pp_wgrad_force : m); }
  // Measured with the wide schedule (tools/bench_wgrad_splits.py, GEMM + slab reduction, T = 8192 / 4096, us; 128x128 kernel first):
  //   768 x 768   (12 tiles): 32.0 | s4 32.2  s6 27.3  s8 25.1  s10 26.0  s12 31.9        / 23.7 | s8 19.9
  //   2304 x 768  (36 tiles): 54.3 | s2 55.8  s3 45.4  s4 54.3  s5 50.0  s7 47.3  s8 67.8 / 35.0 | s3 30.7
  //   768 x 3072  (48 tiles): 84.4 | s2 58.2  s3 66.3  s4 57.7  s5 54.7  s6 80.5          / 52.3 | s2 36.8  s5 39.7
  // More than 256 workgroups is a second round (always worse).  A small output pays for every extra slab (written and read back: 2 x
  // 4 M N bytes per slice), and the K tile of a slice runs faster while fewer CUs compete for L2: its optimum is near 100-110
  // workgroups; the wide FFN outputs want the chip filled.
  const int npn = N % 192 == 0 ? 2 : 1;
  const long tiles = (long)(M / 256) * (N / (96 * npn));
  long s = ((long)M * N < 2000000L ? 112 : 256) / tiles;
  const long nk = K >> 6;
  if (s > nk / 8) s = nk / 8;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  // 129-160 workgroups of the 256 x 192 tile are consistently slower than 96-128 or 180+ (FFN gradients: T = 8192 s3 66.3 us vs s2 58.2 /
  // s4 57.7; T = 1792 s3 31.5 vs s2 26.9; 2304 x 768: s4 54.3 vs s3 45.4 -- cause not found): step down one slice
  if (tiles * s > 128 && tiles * s <= 160 && s > 1) --s;
  if (tiles * s < 64) return 0;      // short token counts: too few K tiles to slice -- the 128x128 kernel's finer tiles fill the chip better
  return (int)s;
}

int gemm_pp_launch_tn(const GemmParams& p, int npn, int splits, hipStream_t s) {
  if (npn == 1) return g_pp_wide ? launch_pp<1, true, true, EPI_SLAB_F32, 0, true>(p, splits, s) : launch_pp<1, true, true, EPI_SLAB_F32>(p, splits, s);
  if (npn == 2) return g_pp_wide ? launch_pp<2, true, true, EPI_SLAB_F32, 0, true>(p, splits, s) : launch_pp<2, true, true, EPI_SLAB_F32>(p, splits, s);
  return set_error(CAREL_ERR_ARG, "gemm_pp_launch_tn: npn = %d not built", npn);
}

// Row-major-A forms with an internal K split (small token counts: 56-112 tiles would leave most CUs idle): `splits` K slices of the
// same tile grid into fp32 slabs [splits][M][N] at p.outf; the caller runs slab_epilogue_kernel over them.  Each slice adds its K tiles
// in the same order as the 128x128 kernel's slices: the same bits.
int gemm_pp_launch_slab(const GemmParams& p, bool bt, int npn, int splits, hipStream_t s) {
  if (!bt) {
    if (npn == 1) return launch_pp<1, false, false, EPI_SLAB_F32, 0, true>(p, splits, s);
    if (npn == 2) return launch_pp<2, false, false, EPI_SLAB_F32, 0, true>(p, splits, s);
  } else {
    if (npn == 1) return launch_pp<1, false, true, EPI_SLAB_F32, 0, true>(p, splits, s);
    if (npn == 2) return launch_pp<2, false, true, EPI_SLAB_F32, 0, true>(p, splits, s);
  }
  return set_error(CAREL_ERR_ARG, "gemm_pp_launch_slab: npn = %d not built", npn);
}

// ---- pair split-K (EXPERIMENTS build only) ---------------------------------------------------------------------------------------------
#ifdef CAREL_EXPERIMENTS
// OFF by default -- built, correct, measured slower (tools/bench_gemm_cold.py, PAIR=0 / 1 on one box, hot / cold operands): FFN2 forward
// 47.1 / 55.4 -> 51.9 / 57.3 us, FFN1 data gradient 45.9 / 53.5 -> 51.5 / 63.8, QKV data gradient 38.4 / 43.8 -> 43.0 / 53.3.  The 256 x 192
// main loop is only ~9 % more efficient per FLOP than the 256 x 96 one (1.40 us per K tile against 2 x 0.77), which saves ~3 us per
// launch; the exchange (196 KiB written through and read back per tile, B idle until A's write lands) and the wider epilogue cost ~9.
static int g_pp_pair = 0;          // tuning hook (carel_gemm_set_variant(200 / 201))
void gemm_pp_pair_enable(int on) { g_pp_pair = on ? 1 : 0; }
// Worth it where the 256 x 96 tiling is L2 -> LDS-bound and the K loop is long enough to pay for the exchange (196 KiB written and read
// per tile): N a multiple of 192 but not a 2304 / 3072-wide output (those already run 256 x 192 / 288 tiles), K >= 1536, and twice the
// 256 x 192 tile count fills the chip in exactly one round.
int gemm_pp_pick_pair(const GemmParams& p, bool bt, int epi) {
  if (!g_pp_pair || !g_pp_wide || !p.pair_flags || !p.splitk_ws || p.resid_stats) return 0;
  if (!(epi == EPI_BIAS_DROP_RESID && !bt) && !(epi == EPI_ADD_F32)) return 0;
  if (p.N != 768 || p.K % 128 || p.K < 1536 || p.M < 1) return 0;
  const long tiles = (long)((p.M + 255) / 256) * (p.N / 192);
  if (tiles * 2 > 256 || tiles * 2 < 224 || (tiles & 7)) return 0;
  if ((size_t)tiles * 256 * 192 * 4 > p.splitk_ws_bytes || tiles * 8 * 4 > (long)PP_PAIR_FLAG_BYTES) return 0;
  return 1;
}
int gemm_pp_launch_pair(const GemmParams& p0, bool bt, int epi, hipStream_t s) {
  if (p0.resid_stats) return set_error(CAREL_ERR_ARG, "gemm_pp_launch_pair: no recomputed-LayerNorm residual on the 256 x 192 pair tiles");
  static std::atomic<unsigned> seq{0};
  GemmParams p = p0;
  unsigned v = seq.fetch_add(1, std::memory_order_relaxed) + 1;
  if (v == 0) v = seq.fetch_add(1, std::memory_order_relaxed) + 1;          // never 0 (the flags start zeroed)
  p.pair_seq = v;
  if (!bt && epi == EPI_BIAS_DROP_RESID) return launch_pp<2, false, false, EPI_BIAS_DROP_RESID, 0, true, true>(p, 2, s);
  if (!bt && epi == EPI_ADD_F32) return launch_pp<2, false, false, EPI_ADD_F32, 0, true, true>(p, 2, s);
  if (bt && epi == EPI_ADD_F32) return launch_pp<2, false, true, EPI_ADD_F32, 0, true, true>(p, 2, s);
  return set_error(CAREL_ERR_ARG, "gemm_pp_launch_pair: unsupported form/epilogue (%d,%d)", (int)bt, epi);
}
#endif   // CAREL_EXPERIMENTS

#ifdef CAREL_GEMM_ABLATE
int gemm_pp_launch_dbg(const GemmParams& p, int npn, int dbg, hipStream_t s) {     // NT, bias -> bf16 epilogue only
#define PPD(N, D) if (npn == N && dbg == D) return g_pp_wide ? launch_pp<N, false, false, EPI_BIAS_BF16, D, true>(p, 1, s) : launch_pp<N, false, false, EPI_BIAS_BF16, D>(p, 1, s)
#define PPDN(N) PPD(N, 1); PPD(N, 2); PPD(N, 3); PPD(N, 4); PPD(N, 5); PPD(N, 6); PPD(N, 7); PPD(N, 8); PPD(N, 9)
  PPDN(1); PPDN(2); PPDN(3);
#undef PPDN
#undef PPD
  return set_error(CAREL_ERR_ARG, "gemm_pp_launch_dbg: (npn, dbg) = (%d, %d) not built", npn, dbg);
}
int gemm_pp_launch_tn_dbg(const GemmParams& p, int npn, int splits, int dbg, hipStream_t s) {     // weight-gradient form, npn 2, wide schedule
#define PPT(D) if (npn == 2 && dbg == D) return launch_pp<2, true, true, EPI_SLAB_F32, D, true>(p, splits, s)
  PPT(1); PPT(2); PPT(3); PPT(4); PPT(6); PPT(7); PPT(8);
#undef PPT
  return set_error(CAREL_ERR_ARG, "gemm_pp_launch_tn_dbg: (npn, dbg) = (%d, %d) not built", npn, dbg);
}
#endif

int gemm_pp_launch(const GemmParams& p, bool bt, int epi, int npn, hipStream_t s) {
  // the wider tiles' epilogues are instantiated WITHOUT the recomputed-LayerNorm residual (no registers for its statistics and gamma / beta):
  // they would add the pre-LayerNorm rows as the residual.  gemm_pp_pick never proposes such a pair; a caller that bypasses it is refused.
  if (p.resid_stats && npn != 1) return set_error(CAREL_ERR_ARG, "gemm_pp_launch: resid_ln_* (recomputed LayerNorm residual) exists on the 96-wide tile only (npn = %d)", npn);
  if (!bt) {
    switch (epi) {
      case EPI_BIAS_BF16: return launch_pp_n<false, EPI_BIAS_BF16>(p, npn, s);
      case EPI_BIAS_GELU: return launch_pp_n<false, EPI_BIAS_GELU>(p, npn, s);
      case EPI_BIAS_GELU_DG: return launch_pp_n<false, EPI_BIAS_GELU_DG>(p, npn, s);
      case EPI_BIAS_DROP_RESID: return launch_pp_n<false, EPI_BIAS_DROP_RESID>(p, npn, s);
      case EPI_ADD_F32: return launch_pp_n<false, EPI_ADD_F32>(p, npn, s);
    }
  } else {
    switch (epi) {
      case EPI_BIAS_BF16: return launch_pp_n<true, EPI_BIAS_BF16>(p, npn, s);
      case EPI_DGELU_BF16: return launch_pp_n<true, EPI_DGELU_BF16>(p, npn, s);
      case EPI_MUL_BF16: return launch_pp_n<true, EPI_MUL_BF16>(p, npn, s);
      case EPI_ADD_F32: return launch_pp_n<true, EPI_ADD_F32>(p, npn, s);
    }
  }
  return set_error(CAREL_ERR_ARG, "gemm_pp_launch: unsupported form/epilogue (%d,%d)", (int)bt, epi);
}

}  // namespace carel
