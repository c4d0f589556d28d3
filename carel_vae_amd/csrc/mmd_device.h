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

}  // namespace carel
