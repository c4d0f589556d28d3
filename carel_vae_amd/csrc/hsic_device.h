// Block-level HSIC (Hilbert-Schmidt independence criterion) forward + backward on samples held in LDS.
// Reference: drl_classifier_ec_hsic.py:529-547 -- K = exp(-D(x)/s_x), L = exp(-D(y)/s_y) with squared
// distances D_ij = |x_i|^2 + |x_j|^2 - 2 x_i.x_j (no eps, no abs), H = I - 11^T/m,
//   HSIC = tr(L H K H) / (m-1)^2 .
// Neither Gram matrix is materialised:  tr(L H K H) = sum_ij L_ij K_ij - (2/m) sum_i RK_i RL_i + TK TL / m^2
// with row sums RK_i = sum_j K_ij, RL_i likewise and totals TK, TL.
#pragma once
#include "carel_common.h"

namespace carel {

struct HsicCfg { int m, d, xs; float inv_sx, inv_sy; };   // xs = LDS row stride (odd)

__device__ __forceinline__ float hsic_kern(const float* X, const float* nrm, int i, int j, int d, int xs, float inv_s) {
  float dot = 0.f;
  for (int k = 0; k < d; ++k) dot = fmaf(X[i * xs + k], X[j * xs + k], dot);
  return __expf(-(nrm[i] + nrm[j] - 2.0f * dot) * inv_s);
}

// All threads call.  X, Y: [m][xs] samples; nx, ny, rk, rl: m floats each (filled here); red >= 64 floats.
// Returns HSIC in every thread; tot[0] = TK, tot[1] = TL are left in red[60], red[61].
__device__ inline float hsic_forward_block(const HsicCfg& c, const float* X, const float* Y, float* nx, float* ny, float* rk,
                                           float* rl, float* red) {
  const int m = c.m;
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < c.d; ++k) { a = fmaf(X[i * c.xs + k], X[i * c.xs + k], a); b = fmaf(Y[i * c.xs + k], Y[i * c.xs + k], b); }
    nx[i] = a; ny[i] = b;
  }
  __syncthreads();
  float slk = 0.f, tk = 0.f, tl = 0.f, cross = 0.f;
  for (int i = threadIdx.x; i < m; i += blockDim.x) {          // one row per thread, fixed order
    float sk = 0.f, sl = 0.f, s2 = 0.f;
    for (int j = 0; j < m; ++j) {
      const float kv = hsic_kern(X, nx, i, j, c.d, c.xs, c.inv_sx), lv = hsic_kern(Y, ny, i, j, c.d, c.xs, c.inv_sy);
      sk += kv; sl += lv; s2 = fmaf(kv, lv, s2);
    }
    rk[i] = sk; rl[i] = sl;
    slk += s2; tk += sk; tl += sl; cross = fmaf(sk, sl, cross);
  }
  slk = block_sum(slk, red); tk = block_sum(tk, red + 16); tl = block_sum(tl, red + 32); cross = block_sum(cross, red + 48);
  __syncthreads();
  if (threadIdx.x == 0) { red[60] = tk; red[61] = tl; }
  __syncthreads();
  const float fm = (float)m;
  return (slk - 2.0f / fm * cross + tk * tl / (fm * fm)) / ((fm - 1.0f) * (fm - 1.0f));
}

// d(gscale * HSIC)/d x_i and /d y_i for row i (one thread per row); gx, gy: [d]
__device__ inline void hsic_backward_row(const HsicCfg& c, const float* X, const float* Y, const float* nx, const float* ny,
                                         const float* rk, const float* rl, float tk, float tl, int i, float gscale, float* gx, float* gy) {
  const int m = c.m;
  const float fm = (float)m, inv = gscale / ((fm - 1.0f) * (fm - 1.0f));
  for (int k = 0; k < c.d; ++k) { gx[k] = 0.f; gy[k] = 0.f; }
  for (int j = 0; j < m; ++j) {
    const float kv = hsic_kern(X, nx, i, j, c.d, c.xs, c.inv_sx), lv = hsic_kern(Y, ny, i, j, c.d, c.xs, c.inv_sy);
    const float lc = lv - (rl[i] + rl[j]) / fm + tl / (fm * fm);       // (H L H)_ij
    const float kc = kv - (rk[i] + rk[j]) / fm + tk / (fm * fm);       // (H K H)_ij
    const float cx = 2.0f * lc * kv * (-2.0f * c.inv_sx) * inv, cy = 2.0f * kc * lv * (-2.0f * c.inv_sy) * inv;
    for (int k = 0; k < c.d; ++k) {
      gx[k] = fmaf(cx, X[i * c.xs + k] - X[j * c.xs + k], gx[k]);
      gy[k] = fmaf(cy, Y[i * c.xs + k] - Y[j * c.xs + k], gy[k]);
    }
  }
}

}  // namespace carel
