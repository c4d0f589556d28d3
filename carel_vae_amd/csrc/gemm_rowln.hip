// Row-band GEMM with the whole sub-layer tail fused:  x = LayerNorm(dropout(A W^T + bias) + residual)   for 768-wide outputs.
// Replaces nn.Linear + nn.Dropout + residual add + nn.LayerNorm of HF BertSelfOutput / BertOutput
// (drl_classifier_ec_mmd_final_mul.py:202-206), i.e. the pair {ping-pong GEMM with the residual epilogue, ln_fwd_kernel} of rounds 1-2.
//
// Why a third GEMM shape.  LayerNorm needs whole 768-wide rows; the 256 x 96 tiles of the ping-pong kernel spread a row over eight
// workgroups, so the pre-LayerNorm sum made a round trip through memory (25 MB written, 25 MB read back by a second kernel,
// 24 launches per step) and -- with ONE round of 256 tiles -- neither the GEMM's 50-MB epilogue nor the LayerNorm pass overlapped with
// any main loop.  Making the eight workgroups of a row band wait for each other would fuse it too, but a kernel whose workgroups
// spin on peers that may not be resident deadlocks as soon as two such kernels share the GPU (two streams, or two processes on one
// device as in tests/test_gpu_dp2.py); a one-directional variant (last workgroup normalises the band) pushes 1.9 MB through one CU.
// So here ONE workgroup owns 32 complete rows:
//   * 8 waves, wave w owns columns [96 w, 96 w + 96): 2 x 6 accumulators of v_mfma_f32_16x16x32_bf16, 48 registers.
//   * the weight rows of a wave are PRIVATE to it, so they never touch the LDS: each lane loads its MFMA operand (one row, 8
//     consecutive k) straight from global memory / L2 with global_load_dwordx4 -- 12 per 64-deep K tile, two tiles in flight (24 KiB
//     per wave, 192 KiB per CU: more than an LDS ring could hold), waited for by the compiler's own in-order vmcnt counts.  No
//     workgroup barrier in the main loop, no LDS-DMA schedule.
//   * the 32 activation rows are shared by all waves: a 768-deep K chunk of them (48 KiB, twelve 32-row ROW images) is staged by
//     LDS-DMA, double-buffered for K > 768, one barrier per chunk.
//   * every workgroup streams the WHOLE weight matrix (1.2 MB at K = 768, 4.7 MB at K = 3072) from L2: the main loop is bound by the
//     L2 -> CU rate -- measured 108 GB/s per CU with the packed weight order below (0.91 us per K tile: 10.9 / 43.7 us), 30-40 GB/s with
//     the nn.Linear layout -- against 8 / 37 us of the ping-pong main loop.
//   * MEASURED (tools/bench_rowln.py, profiles/r03_bench_rowln.txt; hot / cold operands): out-projection 30.1 / 38.7 us against 35.5 / 40.6
//     for GEMM + LayerNorm; FFN2 62.6 / 77.2 against 59.3 / 69.6.  The fixed part is ~19 us: the sub-layer's 87 MB of residual-stream
//     traffic (25 read; 25 + 25 + 12.5 written) at ~5 TB/s -- fusing saves the 25-MB re-read and a launch, not the writes, and with one
//     round of 256 workgroups in lockstep nothing overlaps them.  So the encoder does NOT use this kernel by default (hook 191 / 192);
//     it stays as an operator of the library (carel_gemm_rowln), bit-identical to the two-kernel path.
//   * epilogue: acc + bias goes through the LDS once (the staging buffers are free by then) into ROW layout, and each wave then
//     finishes four rows exactly as ln_fwd_kernel does -- same element-to-lane map, same summation tree (ln_device.h) -- with
//     1-KiB coalesced accesses: residual read, pre-LayerNorm sum written (saved for the backward pass), x f32, x bf16, row statistics.
// Bits: the accumulators see the same MFMA sequence as the ping-pong kernel (same k positions per lane, K tiles in order), the
// dropout / residual expression is the one of epi_out8, the normalisation is ln_normalise: the results equal the two-kernel path bit
// for bit (tests/test_gpu_gemm.py::test_rowln_equals_gemm_then_layernorm_bitwise).
// Only worth it when the row count fills the chip (256 workgroups at T = 8192): packed ECPE batches (~1.8 k rows) keep the old path.
#ifdef CAREL_EXPERIMENTS      // an experiment (built, measured, not adopted): not part of the product library
#include "gemm_epilogue.h"
#include "ln_device.h"

namespace carel {

namespace {

constexpr int RL_ROWS = 32, RL_KC = 768, RL_TILE = RL_ROWS * 128, RL_CHUNK = (RL_KC / 64) * RL_TILE;     // 4 KiB per K tile, 48 KiB per chunk
constexpr int RL_UPITCH = H * 4 + 16;                  // row pitch of the fp32 transpose tile (16 B of padding: the 16 rows of a write land in different banks)
constexpr int RL_LDS = (2 * RL_CHUNK > RL_ROWS * RL_UPITCH) ? 2 * RL_CHUNK : RL_ROWS * RL_UPITCH;

// DBG (timing ablations, wrong results): 1 = no MFMA (the weight stream alone), 2 = no weight loads
// PACKED: the weight in the MFMA-operand order written by pack_rowln_kernel below -- [n / 16][K tile][k32 half][k8 group g][row r][8 k],
// so that the 64 lanes of one load instruction (lane = 16 g + r) read ONE contiguous KiB.  In the nn.Linear layout the sixteen lanes of a
// quarter wave hold sixteen different rows (that IS the operand layout): sixteen 64-byte segments per quarter instead of two 128-byte
// lines, and the vector memory path then delivers 30-40 GB/s per CU instead of ~100 (measured: tools/bench_rowln.py).
template <int DBG, bool PACKED>
__global__ __launch_bounds__(512, 2) void gemm_rowln_kernel(RowLnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long m0 = (long)blockIdx.x * RL_ROWS;
  const int nk = p.K >> 6, nchunk = (nk + 11) / 12;

  // ---- activation rows: chunk c (K tiles 12c .. 12c+11) -> LDS buffer c & 1, six 1-KiB pieces per wave (8 rows x 128 B each) -------
  uint32_t aoff[6]; int adst[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = wave * 6 + i, t = q >> 2, rg = q & 3;
    const int row = rg * 8 + (lane >> 3), c = (lane & 7) ^ (row & 7);
    long rgl = m0 + row; if (rgl > (long)p.M - 1) rgl = (long)p.M - 1;           // rows past M re-read the last row (never stored)
    aoff[i] = (uint32_t)(((rgl - m0) * p.lda + t * 64 + c * 8) * 2);
    adst[i] = t * RL_TILE + rg * 1024;
  }
  const char* a_base = (const char*)(p.A + m0 * p.lda);
  auto stage_chunk = [&](int c) {
    const int tiles = nk - c * 12 < 12 ? nk - c * 12 : 12;
    char* dst = smem + (c & 1) * RL_CHUNK;
    const char* src = a_base + (long)c * RL_KC * 2;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (((wave * 6 + i) >> 2) < tiles)                     // wave-uniform
        __builtin_amdgcn_global_load_lds((const void*)(src + aoff[i]), (CAREL_LDS void*)(dst + adst[i]), 16, 0, 0);
    asm volatile("" ::: "memory");
  };

  // ---- weight operand: lane l holds W[96 w + 16 j + (l & 15)][64 t + 32 s + 8 (l >> 4) .. + 8], straight from global memory -------
  const char* w_lane = PACKED ? (const char*)p.W + (long)(wave * 6) * nk * 2048 + lane * 16
                              : (const char*)p.W + ((long)(wave * 96 + (lane & 15)) * p.ldb + (lane >> 4) * 8) * 2;
  const long w_jstep = PACKED ? (long)nk * 2048 : 16 * p.ldb * 2;
  bf16x8 fb[2][6][2];
  auto load_b = [&](auto ST, int t) {
    constexpr int st = decltype(ST)::value;
    if (DBG == 2 && t > 1) return;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) fb[st][j][s] = *(const bf16x8*)(w_lane + j * w_jstep + (PACKED ? (long)t * 2048 + s * 1024 : (long)t * 128 + s * 64));
  };
  f32x4 acc[2][6];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](auto ST, int t) {
    constexpr int st = decltype(ST)::value;
    const char* tile = smem + ((t / 12) & 1) * RL_CHUNK + (t % 12) * RL_TILE;
    bf16x8 fa[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int s = 0; s < 2; ++s) fa[r][s] = frag16_row(tile, r * 16, s * 32);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          if (DBG == 1) asm volatile("" ::"v"(fb[st][j][s]), "v"(fa[r][s]));
          else acc[r][j] = mfma16(fb[st][j][s], fa[r][s], acc[r][j]);            // swapped operands: D[n][m], 4 consecutive columns per lane
        }
  };

  stage_chunk(0);
  load_b(std::integral_constant<int, 0>{}, 0);
  // chunk 0 staged: everything older than the 12 weight loads just issued has landed (vmcnt retires in issue order)
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // (raw: __syncthreads would also drain the weight loads in flight)
  for (int t = 0; t < nk; t += 2) {                        // nk is even (K multiple of 128, checked by the host)
    if (t % 12 == 0) {
      const int c = t / 12;
      if (c > 0) {
        // chunk c was requested a chunk ago, before every weight load still in flight (the 12 of tile t): in-order retirement
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // ... by every wave; and every wave has finished reading chunk c - 1 (its MFMAs consumed the fragments)
      }
      if (c + 1 < nchunk) stage_chunk(c + 1);               // into the buffer chunk c - 1 occupied
    }
    // (sched_barrier: the machine scheduler otherwise sinks a tile's twelve loads down to just above their first use -- seen in the ISA --
    // and the prefetch distance collapses to nothing)
    load_b(std::integral_constant<int, 1>{}, t + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(std::integral_constant<int, 0>{}, t);
    __builtin_amdgcn_sched_barrier(0);
    // unconditional (past the end: the last tile again, never used): behind a branch the compiler's vmcnt insertion has to assume the
    // loads were NOT issued and waits for the other stage as if it were the newest in flight -- i.e. for this one too
    load_b(std::integral_constant<int, 0>{}, t + 2 < nk ? t + 2 : nk - 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(std::integral_constant<int, 1>{}, t + 1);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue.  The residual rows, gamma and beta are requested first: their latency runs under the transpose through the LDS.
  // Rows past M are clamped for the loads and skipped for the stores.  All four rows of a wave go through every step together
  // (one row at a time, each load -> reduce -> store chain was exposed: 4 x ~3 us).
  const Row G = load_row(p.gamma, lane), Bt = load_row(p.beta, lane);
  Row R[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    long row = m0 + wave * 4 + rr; if (row > (long)p.M - 1) row = (long)p.M - 1;
    R[rr] = load_row(p.resid + row * H, lane);
  }
  // acc + bias -> fp32 transpose tile in the LDS (accumulator layout: row 16 r + (l & 15), columns 96 w + 16 j + 4 (l >> 4) .. + 4)
  __syncthreads();                                           // the staging buffers are free
  {
    const int colq = wave * 96 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float4 b = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) b = *(const float4*)(p.bias + colq + j * 16);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float4 u = {acc[r][j][0] + b.x, acc[r][j][1] + b.y, acc[r][j][2] + b.z, acc[r][j][3] + b.w};
        *(float4*)(smem + (r * 16 + (lane & 15)) * RL_UPITCH + (colq + j * 16) * 4) = u;
      }
    }
  }
  __syncthreads();
  // four rows per wave, the lane map and the arithmetic of ln_fwd_kernel
  Row X[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int lr = wave * 4 + rr;
#pragma unroll
    for (int i = 0; i < NV; ++i) X[rr].v[i] = *(const float4*)(smem + lr * RL_UPITCH + (i * 64 + lane) * 16);
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    long row = m0 + wave * 4 + rr; if (row > (long)p.M - 1) row = (long)p.M - 1;
    const long drow = p.drop_row_map ? (long)p.drop_row_map[row] : row;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const uint32_t e = (uint32_t)(drow * H + (i * 64 + lane) * 4);
      // the expression of epi_out8<EPI_BIAS_DROP_RESID>: v * dropout_mult + r
      float dm[4];
      dropout_mult_n<4>(p.drop, e, dm);
      X[rr].v[i].x = X[rr].v[i].x * dm[0] + R[rr].v[i].x;
      X[rr].v[i].y = X[rr].v[i].y * dm[1] + R[rr].v[i].y;
      X[rr].v[i].z = X[rr].v[i].z * dm[2] + R[rr].v[i].z;
      X[rr].v[i].w = X[rr].v[i].w * dm[3] + R[rr].v[i].w;
    }
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const long row = m0 + wave * 4 + rr;
    if (p.h_out && row < (long)p.M) store_row(p.h_out + row * H, lane, X[rr]);
  }
  float mean[4], rstd[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) ln_normalise(X[rr], G, Bt, p.eps, mean[rr], rstd[rr]);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const long row = m0 + wave * 4 + rr;
    if (row < (long)p.M) {
      if (p.x_f32) store_row(p.x_f32 + row * H, lane, X[rr]);
      if (p.x_bf16) store_row_bf16(p.x_bf16 + row * H, lane, X[rr]);
      if (p.stats && lane == 0) { p.stats[row * 2] = mean[rr]; p.stats[row * 2 + 1] = rstd[rr]; }
    }
  }
}

}  // namespace

static int g_rowln_dbg = 0;          // (ablation builds only)
void gemm_rowln_dbg(int d) { g_rowln_dbg = d; }

// nn.Linear weight [768, K] (leading dimension ldb) -> the packed operand order of gemm_rowln_kernel<.., true>
__global__ __launch_bounds__(256) void pack_rowln_kernel(const bf16_t* __restrict__ W, long ldb, int K, bf16_t* __restrict__ out) {
  const long chunk = (long)blockIdx.x * 256 + threadIdx.x;             // one 16-byte chunk (8 k of one row) per thread, in OUTPUT order
  const int nk = K >> 6;
  if (chunk >= (long)48 * nk * 128) return;
  const int lane = (int)(chunk & 63);
  const long q = chunk >> 6;                                           // (nb, t, s)
  const int s_ = (int)(q & 1), t = (int)((q >> 1) % nk), nb = (int)((q >> 1) / nk);
  const int r = lane & 15, g = lane >> 4;
  *(uint4*)(out + chunk * 8) = *(const uint4*)(W + (long)(nb * 16 + r) * ldb + t * 64 + s_ * 32 + g * 8);
}
int gemm_rowln_pack(const void* W, long ldb, int K, void* out, hipStream_t s) {
  const long chunks = (long)48 * (K >> 6) * 128;
  hipLaunchKernelGGL(pack_rowln_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, (const bf16_t*)W, ldb, K, (bf16_t*)out);
  return check_launch("pack_rowln_kernel");
}

int gemm_rowln_launch(const RowLnParams& p, bool packed, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_rowln_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_rowln_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS);
#ifdef CAREL_GEMM_ABLATE
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_rowln_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_rowln_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS);
#endif
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "gemm_rowln_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr = true;
  }
  const dim3 grid((unsigned)((p.M + RL_ROWS - 1) / RL_ROWS));
#ifdef CAREL_GEMM_ABLATE
  if (g_rowln_dbg == 1 && packed) { hipLaunchKernelGGL((gemm_rowln_kernel<1, true>), grid, dim3(512), RL_LDS, s, p); return check_launch("gemm_rowln_kernel<1>"); }
  if (g_rowln_dbg == 2 && packed) { hipLaunchKernelGGL((gemm_rowln_kernel<2, true>), grid, dim3(512), RL_LDS, s, p); return check_launch("gemm_rowln_kernel<2>"); }
#endif
  if (packed) hipLaunchKernelGGL((gemm_rowln_kernel<0, true>), grid, dim3(512), RL_LDS, s, p);
  else hipLaunchKernelGGL((gemm_rowln_kernel<0, false>), grid, dim3(512), RL_LDS, s, p);
  return check_launch("gemm_rowln_kernel");
}

}  // namespace carel

#endif   // CAREL_EXPERIMENTS
