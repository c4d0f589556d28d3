// Row helpers of the LayerNorm kernels: one wave holds one 768-wide row, 12 floats per lane in three 16-byte chunks (i * 64 + lane) * 4.
// Shared by ln.hip (stand-alone LayerNorm / embeddings) and gemm_rowln.hip (GEMM + bias + dropout + residual + LayerNorm in one
// kernel), so that both produce the same bits: HF BertSelfOutput.LayerNorm / BertOutput.LayerNorm, reached from
// drl_classifier_ec_mmd_final_mul.py:202-206.
#pragma once
#include "carel_common.h"

namespace carel {

constexpr int H = 768;          // hidden size (bert_dim, ref :40)
constexpr int NV = H / 256;     // float4 chunks per lane

struct Row { float4 v[NV]; };

__device__ __forceinline__ Row load_row(const float* __restrict__ p, int lane) {
  Row r;
#pragma unroll
  for (int i = 0; i < NV; ++i) r.v[i] = *(const float4*)(p + (i * 64 + lane) * 4);
  return r;
}
__device__ __forceinline__ void store_row(float* __restrict__ p, int lane, const Row& r) {
#pragma unroll
  for (int i = 0; i < NV; ++i) *(float4*)(p + (i * 64 + lane) * 4) = r.v[i];
}
__device__ __forceinline__ void store_row_bf16(bf16_t* __restrict__ p, int lane, const Row& r) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    uint2 o = {pack2bf(r.v[i].x, r.v[i].y), pack2bf(r.v[i].z, r.v[i].w)};
    *(uint2*)(p + (i * 64 + lane) * 4) = o;
  }
}
__device__ __forceinline__ float row_sum(const Row& r) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (r.v[i].x + r.v[i].y) + (r.v[i].z + r.v[i].w);
  return wave_sum(s);
}
#define ROW_FOREACH(expr)                                             \
  _Pragma("unroll") for (int i = 0; i < NV; ++i) {                    \
    { float& a = A.v[i].x; const float b = B.v[i].x; const float c = Cc.v[i].x; const int e = 0; expr; } \
    { float& a = A.v[i].y; const float b = B.v[i].y; const float c = Cc.v[i].y; const int e = 1; expr; } \
    { float& a = A.v[i].z; const float b = B.v[i].z; const float c = Cc.v[i].z; const int e = 2; expr; } \
    { float& a = A.v[i].w; const float b = B.v[i].w; const float c = Cc.v[i].w; const int e = 3; expr; } \
  }

// normalise X in place given gamma/beta rows; returns mean, rstd
__device__ __forceinline__ void ln_normalise(Row& X, const Row& G, const Row& Bt, float eps, float& mean, float& rstd) {
  mean = row_sum(X) * (1.0f / H);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float a = X.v[i].x - mean, b = X.v[i].y - mean, c = X.v[i].z - mean, d = X.v[i].w - mean;
    s += (a * a + b * b) + (c * c + d * d);
  }
  const float var = wave_sum(s) * (1.0f / H);
  rstd = rsqrtf(var + eps);
  Row& A = X; const Row& B = G; const Row& Cc = Bt;
  ROW_FOREACH(a = ln_apply(a, mean, rstd, b, c); (void)e)
}

}  // namespace carel
