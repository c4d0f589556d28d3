// Block-level RBF-MMD (forward + backward) on samples held in LDS.  Shared by mmd.hip (stand-alone
// MMDStatistic operator) and tail.hip (fused VAE tail).
//
// Reference: MMDStatistic.__call__ drl_classifier_ec_mmd_final_mul.py:547-569, pdist :580-589.
//   d2_ij   = |z_i|^2 + |z_j|^2 - 2 z_i.z_j
//   K_ij    = sum_alpha exp(-alpha * (eps + |d2_ij|))            (sqrt then **2 of the reference folded)
//   mmd     = 2 a01 sum(K12) + a00 (sum(K11) - tr K11) + a11 (sum(K22) - tr K22)
// Z rows 0..n1-1 are sample_1, n1..n1+n2-1 sample_2; LDS row stride `zs` floats (odd => conflict-free).
#pragma once
#include "carel_common.h"

namespace carel {

struct MmdCfg {
  int n1, n2, d, zs;
  int n_alphas;
  float alphas[8];
  float eps;
};

__device__ __forceinline__ float mmd_pair_kernel(const MmdCfg& c, const float* Z, const float* nrm, int i, int j,
                                                 float* dcoef /* sum_a alpha*K_a*sign, may be null */) {
  const float* zi = Z + i * c.zs;
  const float* zj = Z + j * c.zs;
  float dot = 0.f;
  for (int k = 0; k < c.d; ++k) dot = fmaf(zi[k], zj[k], dot);
  const float d2 = nrm[i] + nrm[j] - 2.0f * dot;
  const float ad = c.eps + fabsf(d2);
  float kv = 0.f, dc = 0.f;
  for (int a = 0; a < c.n_alphas; ++a) {
    const float e = expf(-c.alphas[a] * ad);
    kv += e;
    dc = fmaf(c.alphas[a], e, dc);
  }
  if (dcoef) *dcoef = (d2 > 0.f) ? dc : ((d2 < 0.f) ? -dc : 0.f);
  return kv;
}

// All threads of the block call this.  `nrm` (n floats) is filled here.  red: >= 64 floats LDS scratch.
// Returns mmd (valid in every thread).  If kernels_out != null the full Gram matrix is written (ret_matrix).
__device__ inline float mmd_forward_block(const MmdCfg& c, const float* Z, float* nrm, float* red,
                                          float* __restrict__ kernels_out) {
  const int n = c.n1 + c.n2;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < c.d; ++k) s = fmaf(Z[i * c.zs + k], Z[i * c.zs + k], s);
    nrm[i] = s;
  }
  __syncthreads();
  float s11 = 0.f, s22 = 0.f, s12 = 0.f;
  const long total = (long)n * n;
  for (long e = threadIdx.x; e < total; e += blockDim.x) {
    const int i = (int)(e / n), j = (int)(e - (long)i * n);
    const float kv = mmd_pair_kernel(c, Z, nrm, i, j, nullptr);
    if (kernels_out) kernels_out[e] = kv;
    if (i != j) {
      const bool i1 = i < c.n1, j1 = j < c.n1;
      if (i1 && j1) s11 += kv;
      else if (!i1 && !j1) s22 += kv;
      else if (i1 && !j1) s12 += kv;     // K12 block only (the K21 mirror is the factor 2)
    }
  }
  s11 = block_sum(s11, red);
  s22 = block_sum(s22, red + 16);
  s12 = block_sum(s12, red + 32);
  const double a00 = 1.0 / ((double)c.n1 * (c.n1 - 1)), a11 = 1.0 / ((double)c.n2 * (c.n2 - 1));
  const double a01 = -1.0 / ((double)c.n1 * c.n2);
  return (float)(2.0 * a01 * (double)s12 + a00 * (double)s11 + a11 * (double)s22);
}

// Gradient of (gscale * mmd) w.r.t. row i of Z into g[0..d), restricted to partners j = j0, j0+jstep, ...
// (the caller sums the jstep partial results, e.g. with wave shuffles over adjacent lanes).
//   d mmd / d z_i = sum_{j != i} 2 w_ij * (-sum_a alpha K_a sign(d2)) * 2 (z_i - z_j)
__device__ inline void mmd_backward_row(const MmdCfg& c, const float* Z, const float* nrm, int i, float gscale,
                                        float* g /* [d] registers or memory */, int j0 = 0, int jstep = 1) {
  const int n = c.n1 + c.n2;
  const float a00 = (float)(1.0 / ((double)c.n1 * (c.n1 - 1))), a11 = (float)(1.0 / ((double)c.n2 * (c.n2 - 1)));
  const float a01 = (float)(-1.0 / ((double)c.n1 * c.n2));
  for (int k = 0; k < c.d; ++k) g[k] = 0.f;
  const bool i1 = i < c.n1;
  for (int j = j0; j < n; j += jstep) {
    if (j == i) continue;
    float dc;
    (void)mmd_pair_kernel(c, Z, nrm, i, j, &dc);
    const bool j1 = j < c.n1;
    const float w = (i1 == j1) ? (i1 ? a00 : a11) : a01;
    const float coef = -4.0f * w * dc * gscale;
    for (int k = 0; k < c.d; ++k) g[k] = fmaf(coef, Z[i * c.zs + k] - Z[j * c.zs + k], g[k]);
  }
}

// Forward AND backward in one sweep for the case where every row is local (no data-parallel global batch), with the
// feature width known at compile time (everything in registers).  All threads of the block call this; blockDim.x is a
// multiple of 8: 8 adjacent lanes share row i and split its partners j.  Returns mmd and ADDS d(gscale * mmd)/dZ[i] to
// dz[(i % n1) * dz_stride + (i / n1) * CD + k]   (n1 == n2: sample_1 row b and sample_2 row b live in one dz row).
template <int CD>
__device__ inline float mmd_forward_backward_block(const MmdCfg& c, const float* Z, float* nrm, float* red, float gscale,
                                                   float* dz, int dz_stride) {
  const int n = c.n1 + c.n2;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CD; ++k) s = fmaf(Z[i * c.zs + k], Z[i * c.zs + k], s);
    nrm[i] = s;
  }
  __syncthreads();
  const float a00 = (float)(1.0 / ((double)c.n1 * (c.n1 - 1))), a11 = (float)(1.0 / ((double)c.n2 * (c.n2 - 1)));
  const float a01 = (float)(-1.0 / ((double)c.n1 * c.n2));
  float s11 = 0.f, s22 = 0.f, s12 = 0.f;
  const int grp = threadIdx.x & 7;
  for (int base = 0; base < n; base += blockDim.x / 8) {
    const int i = base + (threadIdx.x >> 3);
    const bool live = i < n;
    const int ii = live ? i : 0;
    float zi[CD], g[CD];
#pragma unroll
    for (int k = 0; k < CD; ++k) { zi[k] = Z[ii * c.zs + k]; g[k] = 0.f; }
    const float ni = nrm[ii];
    const bool i1 = ii < c.n1;
    if (live) {
      for (int j = grp; j < n; j += 8) {
        if (j == i) continue;
        float zj[CD];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < CD; ++k) { zj[k] = Z[j * c.zs + k]; dot = fmaf(zi[k], zj[k], dot); }
        const float d2 = ni + nrm[j] - 2.0f * dot;
        const float ad = c.eps + fabsf(d2);
        float kv = 0.f, dc = 0.f;
        for (int al = 0; al < c.n_alphas; ++al) {
          const float e = expf(-c.alphas[al] * ad);
          kv += e;
          dc = fmaf(c.alphas[al], e, dc);
        }
        const float dcs = (d2 > 0.f) ? dc : ((d2 < 0.f) ? -dc : 0.f);
        const bool j1 = j < c.n1;
        const float w = (i1 == j1) ? (i1 ? a00 : a11) : a01;
        const float coef = -4.0f * w * dcs * gscale;
#pragma unroll
        for (int k = 0; k < CD; ++k) g[k] = fmaf(coef, zi[k] - zj[k], g[k]);
        if (i1 && j1) s11 += kv;
        else if (!i1 && !j1) s22 += kv;
        else if (i1 && !j1) s12 += kv;        // K12 block only (the K21 mirror is the factor 2)
      }
    }
#pragma unroll
    for (int k = 0; k < CD; ++k) {
      float v = g[k];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
      if (live && grp == 0) dz[(i % c.n1) * dz_stride + (i / c.n1) * CD + k] += v;
    }
  }
  s11 = block_sum(s11, red);
  s22 = block_sum(s22, red + 16);
  s12 = block_sum(s12, red + 32);
  const double b00 = 1.0 / ((double)c.n1 * (c.n1 - 1)), b11 = 1.0 / ((double)c.n2 * (c.n2 - 1));
  const double b01 = -1.0 / ((double)c.n1 * c.n2);
  return (float)(2.0 * b01 * (double)s12 + b00 * (double)s11 + b11 * (double)s22);
}

}  // namespace carel
