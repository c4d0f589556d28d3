// Shared device helpers for the CAREL-VAE gfx950 kernels: vector types, bf16 conversion, the
// counter-based dropout hash (bit-identical to oracle/carel_oracle.py::_mix32), wave reductions and
// the LDS tile images + MFMA fragment loaders used by gemm.hip / attention.hip.
//
// gfx950 only (wave64, MFMA 16x16x32 / 32x32x16 bf16, ds_read_b64_tr_b16, global_load_lds_dwordx4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace carel {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;   // raw bf16 bits in memory

#define CAREL_LDS __attribute__((address_space(3)))

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {   // round-to-nearest-even, NaN stays NaN
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
// two values per v_cvt_pk_bf16_f32 (written as two scalar conversions + shift + or, the compiler emits one conversion per value and
// two more instructions to merge them: 4 VALU per pair instead of 1 -- every bf16 store of the library goes through here)
typedef __attribute__((ext_vector_type(2))) float f32x2_cvt;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_cvt v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// ---------------------------------------------------------------- dropout hash
__device__ __host__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
struct Dropout {          // passed by value to kernels
  uint32_t key;           // mix32(seed + site * 0x9E3779B9)
  uint32_t thresh;        // keep iff hash >= thresh ; 0 => dropout disabled
  float scale;            // 1/(1-p)
  uint32_t idx_offset;    // added to the linear element index (DP shard offset)
};
// Element j = idx + idx_offset is kept iff the 16-bit half (j & 1) of mix32((j >> 1) ^ key) is >= the upper 16 bits of the threshold:
// two consecutive elements share one hash (two quarter-rate v_mul_lo_u32 each; the attention kernels take 25 M decisions per layer and
// direction and were spending ~4 400 cycles per wave on them).  dropout_mult = one element; dropout_hash2 + dropout_pick = the pair.
__device__ __forceinline__ float dropout_pick(const Dropout& d, uint32_t h, uint32_t odd) {
  return ((odd ? (h >> 16) : (h & 0xffffu)) >= (d.thresh >> 16)) ? d.scale : 0.0f;
}
__device__ __forceinline__ uint32_t dropout_hash2(const Dropout& d, uint32_t idx) {      // hash of the pair that holds element idx
  return mix32(((idx + d.idx_offset) >> 1) ^ d.key);
}
__device__ __forceinline__ float dropout_mult(const Dropout& d, uint32_t idx) {
  if (d.thresh == 0u) return 1.0f;
  return dropout_pick(d, dropout_hash2(d, idx), (idx + d.idx_offset) & 1u);
}
// The multipliers of the N (even) consecutive elements idx .. idx + N - 1, the same values as N calls of dropout_mult.  When the first element
// is the even half of its pair (every caller: idx a multiple of 4 or 8, the shard offset even) they are N/2 whole pairs: N/2 hashes and one test
// of the rate for all of them.  (Element by element the epilogues compiled to one full hash -- two quarter-rate v_mul_lo_u32 -- and one
// branch on the rate PER ELEMENT: 48 hashes per lane in the 256 x 96 tile's residual epilogue.)
template <int N>
__device__ __forceinline__ void dropout_mult_n(const Dropout& d, uint32_t idx, float* m) {
  static_assert(N % 2 == 0, "whole pairs");
  if (d.thresh == 0u) {
#pragma unroll
    for (int i = 0; i < N; ++i) m[i] = 1.0f;
    return;
  }
  const uint32_t j0 = idx + d.idx_offset;
  if ((j0 & 1u) == 0u) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const uint32_t h = mix32(((j0 >> 1) + (uint32_t)i) ^ d.key);
      m[2 * i] = dropout_pick(d, h, 0u); m[2 * i + 1] = dropout_pick(d, h, 1u);
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) m[i] = dropout_pick(d, dropout_hash2(d, idx + (uint32_t)i), (j0 + (uint32_t)i) & 1u);
  }
}

// One element of a LayerNorm output from its input, the row's statistics and the column's gamma / beta: THE expression of the LayerNorm
// kernels (ln_device.h: ln_normalise) -- also used by the GEMM epilogues that recompute a residual LN(h) from the saved pre-LayerNorm
// rows instead of reading a stored copy (gemm_epilogue.h: resid_stats), so that both give the same bits.
__device__ __forceinline__ float ln_apply(float a, float mean, float rstd, float g, float b) { return (a - mean) * rstd * g + b; }

// ---------------------------------------------------------------- reductions (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over each aligned group of 16 lanes, result in all 16: four DPP adds (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror,
// row_mirror) instead of four ds_bpermute round trips.  Every stage adds two group sums that are uniform within their groups, so the
// result has the bits of the xor-butterfly (1, 2, 4, 8) it replaces.
__device__ __forceinline__ float row16_sum(float v) {
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true)); x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true)); x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true)); x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x <= 1024; `red` is >= 16 floats of LDS; result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// ---------------------------------------------------------------- math
// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32-level): one exp, one reciprocal, a
// degree-5 Horner chain -- about a third of the instructions of the correctly-rounded erff, which matters in
// the GEMM epilogues that evaluate it 25 M times per layer.  ez = exp(-z^2) is returned for re-use.
__device__ __forceinline__ float erf_as(float z, float& ez) {
  const float az = fabsf(z);
  const float t = __frcp_rn(fmaf(0.3275911f, az, 1.0f));
  ez = __expf(-az * az);
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = 1.0f - poly * t * ez;
  return copysignf(e, z);
}
// GELU(x) = x * Phi(x), exact-erf form (HF `gelu`), and its derivative Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf(float x) {
  float ez;
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f, ez));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float ez;                                     // = exp(-x^2 / 2)
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f, ez));
  return fmaf(x * 0.39894228040143268f, ez, cdf);
}

// Two elements at a time on the packed-fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of fp32 per
// instruction): the same operations in the same order as erf_as / gelu_erf / gelu_erf_grad, so the results are those of the
// scalar functions; only the reciprocal and the exponential stay one per element.  The GEMM epilogues that evaluate GELU
// 25 M times per layer are VALU-bound on exactly this arithmetic.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 erf_as2(f32x2 z, f32x2& ez) {
  const f32x2 az = {fabsf(z.x), fabsf(z.y)};
  const f32x2 den = pk_fma(pk_splat(0.3275911f), az, pk_splat(1.0f));
  const f32x2 t = {__frcp_rn(den.x), __frcp_rn(den.y)};
  const f32x2 m = -az * az;
  ez = f32x2{__expf(m.x), __expf(m.y)};
  f32x2 poly = pk_fma(pk_splat(1.061405429f), t, pk_splat(-1.453152027f));
  poly = pk_fma(poly, t, pk_splat(1.421413741f));
  poly = pk_fma(poly, t, pk_splat(-0.284496736f));
  poly = pk_fma(poly, t, pk_splat(0.254829592f));
  const f32x2 e = pk_splat(1.0f) - poly * t * ez;
  return f32x2{copysignf(e.x, z.x), copysignf(e.y, z.y)};
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  f32x2 ez;
  return pk_splat(0.5f) * x * (pk_splat(1.0f) + erf_as2(x * pk_splat(0.70710678118654752f), ez));
}
// both at once (one erf, one exp): g = gelu(x), dg = gelu'(x); the same operations as the two functions above
__device__ __forceinline__ void gelu_erf_both2(f32x2 x, f32x2& g, f32x2& dg) {
  f32x2 ez;
  const f32x2 one_p = pk_splat(1.0f) + erf_as2(x * pk_splat(0.70710678118654752f), ez);
  g = pk_splat(0.5f) * x * one_p;
  dg = pk_fma(x * pk_splat(0.39894228040143268f), ez, pk_splat(0.5f) * one_p);
}
__device__ __forceinline__ f32x2 gelu_erf_grad2(f32x2 x) {
  f32x2 ez;
  const f32x2 cdf = pk_splat(0.5f) * (pk_splat(1.0f) + erf_as2(x * pk_splat(0.70710678118654752f), ez));
  return pk_fma(x * pk_splat(0.39894228040143268f), ez, cdf);
}

// =========================================================================================
// LDS tile images.  Both are written by global_load_lds_dwordx4 (lane-linear 1 KiB per wave
// instruction), so the XOR swizzle is applied to the per-lane SOURCE address and again on reads.
//
//  ROW image  ("K-contiguous"):  [rows][64 bf16]   128-B rows, 8 chunks of 16 B per row.
//      physical chunk = chunk ^ (row & 7)          -> ds_read_b128 fragment reads conflict-free
//  COL image  ("K-strided", read with ds_read_b64_tr_b16): [64 k-rows][128 bf16]  256-B rows,
//      16 chunks per row.  physical chunk = chunk ^ swz_col(krow)
// =========================================================================================
__device__ __forceinline__ int swz_col(int krow) { return ((krow & 3) << 1) | (((krow >> 3) & 1) << 3); }

// byte offset inside a ROW image of (row, 16-B chunk)
__device__ __forceinline__ int row_img_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// byte offset inside a COL image of (krow, 16-B chunk)
__device__ __forceinline__ int col_img_off(int krow, int chunk) { return krow * 256 + ((chunk ^ swz_col(krow)) << 4); }

// Issue the glds loads that fill a ROW image tile of `ROWS` rows x 64 bf16 from a row-major global
// matrix (leading dimension ld elements), starting at element (row0, k0).  256 threads.
template <int ROWS>
__device__ __forceinline__ void stage_row_image(const bf16_t* __restrict__ g, long ld, long row0, long k0,
                                                char* lds_tile) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NINSTR = ROWS / 8;            // 1 KiB pieces (8 rows each)
#pragma unroll
  for (int i = 0; i < NINSTR / 4; ++i) {
    const int q = wave * (NINSTR / 4) + i;
    const int r = q * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7);
    const bf16_t* src = g + (row0 + r) * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds(src, (CAREL_LDS void*)(lds_tile + q * 1024), 16, 0, 0);
  }
}
// The same two stagings with the address split into a per-lane 32-bit byte offset that is computed ONCE per kernel
// (row_lane_off / col_lane_off, one per DMA instruction of the wave) and a wave-uniform byte pointer that the caller
// advances per K step: the loads then take the SGPR-base + VGPR-offset form and the loop body carries no 64-bit
// multiply-adds (the plain forms above cost ~60 VALU instructions per K step for eight loads).
__device__ __forceinline__ uint32_t row_lane_off(long ld, int i) {            // 128-row ROW image, DMA i of this wave (0..3)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = wave * 4 + i, r = q * 8 + (lane >> 3), c = (lane & 7) ^ (r & 7);
  return (uint32_t)((r * ld + c * 8) * 2);
}
__device__ __forceinline__ uint32_t col_lane_off(long ld, int i) {            // COL image, DMA i of this wave (0..3)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = wave * 4 + i, r = q * 4 + (lane >> 4), c = (lane & 15) ^ swz_col(r);
  return (uint32_t)((r * ld + c * 8) * 2);
}
// base: wave-uniform byte pointer to element (row0, k0) [ROW] / (krow0, x0) [COL] of the global matrix
__device__ __forceinline__ void stage_image_fast(const char* base, const uint32_t (&off)[4], char* lds_tile) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_global_load_lds((const void*)(base + off[i]), (CAREL_LDS void*)(lds_tile + (wave * 4 + i) * 1024), 16, 0, 0);
}
// COL image: 64 k-rows x 128 columns, global matrix is [k][x] row-major.
__device__ __forceinline__ void stage_col_image(const bf16_t* __restrict__ g, long ld, long krow0, long x0,
                                                char* lds_tile) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = wave * 4 + i;               // 16 pieces of 4 k-rows
    const int r = q * 4 + (lane >> 4);
    const int c = (lane & 15) ^ swz_col(r);
    const bf16_t* src = g + (krow0 + r) * ld + x0 + c * 8;
    __builtin_amdgcn_global_load_lds(src, (CAREL_LDS void*)(lds_tile + q * 1024), 16, 0, 0);
  }
}

// ---- 16x16x32 fragments: lane l holds X[row = x0 + (l&15)][k = k0 + 8*(l>>4) + j], j = 0..7 ----
// from a ROW image (k0 multiple of 8 within the 64-wide tile)
__device__ __forceinline__ bf16x8 frag16_row(const char* tile, int x0, int k0) {
  const int l = threadIdx.x & 63;
  const int row = x0 + (l & 15), chunk = (k0 >> 3) + (l >> 4);
  return *(const bf16x8*)(tile + row_img_off(row, chunk));
}
// from a COL image: two transposed reads of 4 k-rows x 16 columns per 16-lane group
__device__ __forceinline__ bf16x8 frag16_col(const char* tile, int x0, int k0) {
  const int l = threadIdx.x & 63;
  const int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  const int chunk = (x0 >> 3) + (p >> 1), sub = (p & 1) * 8;
  const int r0 = k0 + 8 * g + q, r1 = r0 + 4;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((CAREL_LDS s16x4*)(tile + col_img_off(r0, chunk) + sub));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((CAREL_LDS s16x4*)(tile + col_img_off(r1, chunk) + sub));
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// 32x32 accumulator: register r (0..15) of lane l is element [row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
__device__ __forceinline__ int acc32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace carel
