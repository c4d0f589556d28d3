// VI / CLUB disentanglement head of the ablation script drl_classifier_ec_vi.py:
//   approximation network p(e|c):  mu_hat = W2 relu(W1 c + b1) + b2 ;  lv_hat = tanh(V2 relu(V1 c + d1) + d2)   (:156-163, :343-348)
//   get_ec_aprx_loss (:422-427):   -mean_b sum_d( -(mu_hat - e)^2 / exp(lv_hat) - lv_hat )   on c.detach(); trains the net only
//   get_ec_upper_loss (:429-440):  mean_b sum_d( -(mu_hat - e_b)^2 + (mu_hat - e_perm(b))^2 ) / exp(lv_hat) / 2   (CLUB bound)
// z = [e | c] are the sampled emotion / cause embeddings [B, 2D], D <= 32.  One workgroup; thread per sample for the
// forward and the per-sample backward, thread per weight for the parameter gradients (fixed summation order).
#include "carel_hip_internal.h"

namespace carel {

struct ViNet { const float* w1; const float* b1; const float* w2; const float* b2; };
struct ViArgs {
  const float* z; int B, D;
  ViNet mu, lv;
  const int* perm;
  float* loss;
  float* g[8];           // aprx: d loss / d {mu.w1, mu.b1, mu.w2, mu.b2, lv.w1, lv.b1, lv.w2, lv.b2}
  float* dz;             // upper: d loss / d z  [B, 2D]
};

// h = relu(W1 c + b1) [D], out = W2 h + b2 [D]
__device__ __forceinline__ void mlp_fwd(const ViNet& n, const float* c, int D, float* h, float* out) {
  for (int o = 0; o < D; ++o) {
    float s = n.b1[o];
    for (int i = 0; i < D; ++i) s = fmaf(n.w1[o * D + i], c[i], s);
    h[o] = fmaxf(s, 0.f);
  }
  for (int o = 0; o < D; ++o) {
    float s = n.b2[o];
    for (int i = 0; i < D; ++i) s = fmaf(n.w2[o * D + i], h[i], s);
    out[o] = s;
  }
}
// given dout [D] (gradient at the second linear's output): dh = relu'(h) * W2^T dout ; dc += W1^T dh
__device__ __forceinline__ void mlp_bwd(const ViNet& n, const float* h, const float* dout, int D, float* dh, float* dc) {
  for (int i = 0; i < D; ++i) {
    float s = 0.f;
    for (int o = 0; o < D; ++o) s = fmaf(n.w2[o * D + i], dout[o], s);
    dh[i] = h[i] > 0.f ? s : 0.f;
  }
  if (dc) {
    for (int i = 0; i < D; ++i) {
      float s = 0.f;
      for (int o = 0; o < D; ++o) s = fmaf(n.w1[o * D + i], dh[o], s);
      dc[i] += s;
    }
  }
}

template <int UPPER>
__global__ __launch_bounds__(256) void vi_kernel(ViArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int B = a.B, D = a.D, t = threadIdx.x;
  float* red = (float*)smem_raw;            // 16
  float* hm = red + 16;                     // [B][D] hidden of the mu net
  float* hl = hm + B * D;                   // [B][D] hidden of the log-var net
  float* dom = hl + B * D;                  // [B][D] d loss / d mu_hat
  float* dol = dom + B * D;                 // [B][D] d loss / d (pre-tanh log-var output)
  float* dhm = dol + B * D;                 // [B][D]
  float* dhl = dhm + B * D;                 // [B][D]
  float part = 0.f;
  for (int b = t; b < B; b += blockDim.x) {
    const float* e = a.z + (long)b * 2 * D;
    const float* c = e + D;
    float mu[32], lp[32];
    mlp_fwd(a.mu, c, D, hm + b * D, mu);
    mlp_fwd(a.lv, c, D, hl + b * D, lp);
    const float* e2 = UPPER ? a.z + (long)a.perm[b] * 2 * D : e;
    float dc[32];
    for (int d = 0; d < D; ++d) dc[d] = 0.f;
    for (int d = 0; d < D; ++d) {
      const float lv = tanhf(lp[d]);
      const float s = expf(-lv);
      const float r = mu[d] - e[d];
      float dmu, dlv;
      if (UPPER) {
        const float r2 = mu[d] - e2[d];
        part += (-r * r + r2 * r2) * s;
        const float k = 0.5f / B;
        dmu = k * (-2.f * r + 2.f * r2) * s;
        dlv = k * (r * r - r2 * r2) * s;
        atomicAdd(&a.dz[(long)b * 2 * D + d], k * 2.f * r * s);                 // d/d e_b      (positive term)
        atomicAdd(&a.dz[(long)a.perm[b] * 2 * D + d], -k * 2.f * r2 * s);       // d/d e_perm(b) (negative term)
      } else {
        part += r * r * s + lv;                                                  // = -(log-likelihood summand)
        dmu = 2.f * r * s / B;
        dlv = (-r * r * s + 1.f) / B;
      }
      dom[b * D + d] = dmu;
      dol[b * D + d] = dlv * (1.f - lv * lv);                                    // through tanh
    }
    mlp_bwd(a.mu, hm + b * D, dom + b * D, D, dhm + b * D, UPPER ? dc : nullptr);
    mlp_bwd(a.lv, hl + b * D, dol + b * D, D, dhl + b * D, UPPER ? dc : nullptr);
    if (UPPER) for (int d = 0; d < D; ++d) atomicAdd(&a.dz[(long)b * 2 * D + D + d], dc[d]);   // d/d c_b through both nets
  }
  part = block_sum(part, red);
  if (t == 0) a.loss[0] = UPPER ? part * 0.5f / B : part / B;
  if (UPPER) return;
  __syncthreads();
  // parameter gradients of the aprx loss: thread per element, loop over the batch
  const int nw = D * D;
  for (int e = t; e < 4 * nw + 4 * D; e += blockDim.x) {
    float s = 0.f;
    if (e < 4 * nw) {
      const int which = e / nw, r = e - which * nw, o = r / D, i = r - o * D;
      // which: 0 mu.w1 (dh_mu x c), 1 mu.w2 (dout_mu x h_mu), 2 lv.w1, 3 lv.w2
      for (int b = 0; b < B; ++b) {
        const float* c = a.z + (long)b * 2 * D + D;
        if (which == 0) s = fmaf(dhm[b * D + o], c[i], s);
        else if (which == 1) s = fmaf(dom[b * D + o], hm[b * D + i], s);
        else if (which == 2) s = fmaf(dhl[b * D + o], c[i], s);
        else s = fmaf(dol[b * D + o], hl[b * D + i], s);
      }
      a.g[which == 0 ? 0 : which == 1 ? 2 : which == 2 ? 4 : 6][r] = s;
    } else {
      const int r = e - 4 * nw, which = r / D, o = r - which * D;
      for (int b = 0; b < B; ++b) s += (which == 0 ? dhm : which == 1 ? dom : which == 2 ? dhl : dol)[b * D + o];
      a.g[which == 0 ? 1 : which == 1 ? 3 : which == 2 ? 5 : 7][o] = s;
    }
  }
}

}  // namespace carel

using namespace carel;

static int vi_launch(const carel_vi_args* a, int upper, hipStream_t stream, const char* who) {
  if (!a || !a->z || !a->loss_out) return set_error(CAREL_ERR_ARG, "%s: null tensor", who);
  if (a->batch < 1 || a->ec_dim < 1 || a->ec_dim > 32) return set_error(CAREL_ERR_SHAPE, "%s: need batch >= 1 and ec_dim <= 32", who);
  for (int i = 0; i < 8; ++i) if (!a->net[i]) return set_error(CAREL_ERR_ARG, "%s: null approximation-network tensor", who);
  ViArgs k;
  k.z = (const float*)a->z; k.B = a->batch; k.D = a->ec_dim;
  k.mu = ViNet{(const float*)a->net[0], (const float*)a->net[1], (const float*)a->net[2], (const float*)a->net[3]};
  k.lv = ViNet{(const float*)a->net[4], (const float*)a->net[5], (const float*)a->net[6], (const float*)a->net[7]};
  k.perm = (const int*)a->perm; k.loss = (float*)a->loss_out; k.dz = (float*)a->dz;
  for (int i = 0; i < 8; ++i) k.g[i] = (float*)a->d_net[i];
  const size_t lds = sizeof(float) * (16 + 6 * (size_t)a->batch * a->ec_dim);
  if (lds > 160 * 1024) return set_error(CAREL_ERR_SHAPE, "%s: batch too large for one workgroup", who);
  if (upper) {
    if (!a->perm || !a->dz) return set_error(CAREL_ERR_ARG, "%s: perm and dz are required", who);
    hipError_t e = hipMemsetAsync(a->dz, 0, (size_t)a->batch * 2 * a->ec_dim * sizeof(float), stream);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "%s: memset: %s", who, hipGetErrorString(e));
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)vi_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return set_error(CAREL_ERR_HIP, "%s: hipFuncSetAttribute failed", who);
    hipLaunchKernelGGL(vi_kernel<1>, dim3(1), dim3(256), lds, stream, k);
  } else {
    for (int i = 0; i < 8; ++i) if (!a->d_net[i]) return set_error(CAREL_ERR_ARG, "%s: null gradient tensor", who);
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)vi_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return set_error(CAREL_ERR_HIP, "%s: hipFuncSetAttribute failed", who);
    hipLaunchKernelGGL(vi_kernel<0>, dim3(1), dim3(256), lds, stream, k);
  }
  return check_launch("vi_kernel");
}

extern "C" int carel_vi_aprx(const carel_vi_args* a, void* stream) { return vi_launch(a, 0, (hipStream_t)stream, "carel_vi_aprx"); }
extern "C" int carel_vi_upper(const carel_vi_args* a, void* stream) { return vi_launch(a, 1, (hipStream_t)stream, "carel_vi_upper"); }
