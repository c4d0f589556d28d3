// Fused Adam over a flat fp32 parameter buffer + bf16 shadow refresh.  Replaces
// torch.optim.Adam(model.get_params(), lr).step() (drl_classifier_ec_mmd_final_mul.py:936, :842): defaults
// betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad, dense (rows with zero gradient still move).
// HBM-bound: 16 B read + 12 B written per parameter (+2 B for the bf16 copy the MFMA GEMMs consume).
#include "carel_hip_internal.h"

namespace carel {

struct AdamArgs {
  float* p; const float* g; float* m; float* v; bf16_t* shadow;
  long n;
  float lr_over_bc1, inv_sqrt_bc2, b1, b2, eps, grad_scale;
  long skip_lo, skip_hi; const float* skip_flag;     // [skip_lo, skip_hi) untouched when *skip_flag != 0
  const float* skip_count;                           // optional device scalar: steps the skip range has been frozen so far
  float lr, step;                                    //   -> its bias corrections use (step - *skip_count), like torch's per-parameter step
  const float* grad_scale_dev;                       // optional device scalar multiplied into the gradient (clip coefficient)
  float decay;                                       // AdamW: p *= decay (= 1 - lr * weight_decay) first, inside the decay segments
  const long* seg; int nseg;                         // sorted [start, end) pairs (elements, multiples of 4) that take the decay
};
// is element i (a multiple of 4) inside one of the sorted decay segments?
__device__ __forceinline__ bool adam_decays(const AdamArgs& a, long i) {
  int lo = 0, hi = a.nseg;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.seg[2 * mid + 1] <= i) lo = mid + 1; else hi = mid; }
  return lo < a.nseg && a.seg[2 * lo] <= i;
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a) {
  g *= a.grad_scale;
  m = m + (1.0f - a.b1) * (g - m);                       // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.b2 + (1.0f - a.b2) * g * g;                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;  // (sqrt(v) / sqrt(bc2)).add_(eps)
  p = p - a.lr_over_bc1 * (m / denom);                   // param.addcdiv_(exp_avg, denom, value=-lr/bc1)
}

// torch.optim.Adam keeps one step counter PER PARAMETER and does not advance it when the parameter's grad is None: the range
// that can be frozen (the pair head) therefore has its own bias corrections once it has been frozen at least once
__device__ __forceinline__ AdamArgs adam_range_args(const AdamArgs& a) {
  AdamArgs r = a;
  const float own = a.step - a.skip_count[0];
  r.lr_over_bc1 = a.lr / (1.0f - powf(a.b1, own));
  r.inv_sqrt_bc2 = 1.0f / sqrtf(1.0f - powf(a.b2, own));
  return r;
}

__global__ void adam_skip_bump_kernel(const float* __restrict__ flag, float* __restrict__ count) {
  if (flag[0] != 0.f) count[0] += 1.0f;
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  const bool skipping = a.skip_flag && a.skip_flag[0] != 0.f;
  if (a.grad_scale_dev) a.grad_scale *= a.grad_scale_dev[0];
  const bool own_steps = a.skip_count && a.skip_count[0] != 0.f && a.skip_hi > a.skip_lo;
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i < a.n; i += stride) {
    if (i + 4 <= a.n) {
      float4 p = *(float4*)(a.p + i);
      if (a.nseg > 0 && adam_decays(a, i)) { p.x *= a.decay; p.y *= a.decay; p.z *= a.decay; p.w *= a.decay; }   // param.mul_(1 - lr * weight_decay)
      const float4 g = *(const float4*)(a.g + i);
      float4 m = *(float4*)(a.m + i), v = *(float4*)(a.v + i);
      const bool s0 = skipping && i + 0 >= a.skip_lo && i + 0 < a.skip_hi, s1 = skipping && i + 1 >= a.skip_lo && i + 1 < a.skip_hi;
      const bool s2 = skipping && i + 2 >= a.skip_lo && i + 2 < a.skip_hi, s3 = skipping && i + 3 >= a.skip_lo && i + 3 < a.skip_hi;
      if (own_steps && i + 3 >= a.skip_lo && i < a.skip_hi) {       // (rare: the few float4 groups that touch the pair head)
        const AdamArgs r = adam_range_args(a);
        const bool r0 = i + 0 >= a.skip_lo && i + 0 < a.skip_hi, r1 = i + 1 >= a.skip_lo && i + 1 < a.skip_hi;
        const bool r2 = i + 2 >= a.skip_lo && i + 2 < a.skip_hi, r3 = i + 3 >= a.skip_lo && i + 3 < a.skip_hi;
        if (!s0) adam1(p.x, g.x, m.x, v.x, r0 ? r : a);
        if (!s1) adam1(p.y, g.y, m.y, v.y, r1 ? r : a);
        if (!s2) adam1(p.z, g.z, m.z, v.z, r2 ? r : a);
        if (!s3) adam1(p.w, g.w, m.w, v.w, r3 ? r : a);
      } else {
        if (!s0) adam1(p.x, g.x, m.x, v.x, a);
        if (!s1) adam1(p.y, g.y, m.y, v.y, a);
        if (!s2) adam1(p.z, g.z, m.z, v.z, a);
        if (!s3) adam1(p.w, g.w, m.w, v.w, a);
      }
      *(float4*)(a.p + i) = p; *(float4*)(a.m + i) = m; *(float4*)(a.v + i) = v;
      if (a.shadow) { uint2 o = {pack2bf(p.x, p.y), pack2bf(p.z, p.w)}; *(uint2*)(a.shadow + i) = o; }
    } else {
      for (long j = i; j < a.n; ++j) {
        if (skipping && j >= a.skip_lo && j < a.skip_hi) continue;
        float p = a.p[j], m = a.m[j], v = a.v[j];
        if (a.nseg > 0 && adam_decays(a, j & ~3L)) p *= a.decay;
        if (own_steps && j >= a.skip_lo && j < a.skip_hi) adam1(p, a.g[j], m, v, adam_range_args(a));
        else adam1(p, a.g[j], m, v, a);
        a.p[j] = p; a.m[j] = m; a.v[j] = v;
        if (a.shadow) a.shadow[j] = f2bf(p);
      }
    }
  }
}

// torch.optim.RMSprop defaults (drl_classifier_en.py:1056-1060: the five discriminator optimisers): alpha 0.99, eps 1e-8,
// no momentum, not centered, no weight decay:  v = alpha v + (1 - alpha) g^2 ;  p -= lr g / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v, long n,
                                                      float lr, float alpha, float eps) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float gi = g[i];
    const float vi = v[i] * alpha + (1.0f - alpha) * gi * gi;        // square_avg.mul_(alpha).addcmul_(grad, grad, 1 - alpha)
    v[i] = vi;
    p[i] = p[i] - lr * (gi / (sqrtf(vi) + eps));                     // param.addcdiv_(grad, sqrt(square_avg) + eps, value=-lr)
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    if (i + 4 <= n) {
      const float4 p = *(const float4*)(src + i);
      uint2 o = {pack2bf(p.x, p.y), pack2bf(p.z, p.w)};
      *(uint2*)(dst + i) = o;
    } else {
      for (long j = i; j < n; ++j) dst[j] = f2bf(src[j]);
    }
  }
}

}  // namespace carel

using namespace carel;

// Grid of the Adam launch (tuning hook 280 + k in the experiments build: cap = 32 << k workgroups, 292 = none): one float4 per thread up to 134 M
// elements streams fastest stand-alone (tools/ubench/adam_stream.hip: a capped, grid-striding launch is 8-14 % slower) ...
CAREL_TUNABLE(long, g_adam_grid_cap, 131072);
#ifdef CAREL_EXPERIMENTS
namespace carel { void adam_grid_cap(long cap) { g_adam_grid_cap = cap; } }
#endif
static unsigned grid_for(long n, long cap = 131072) {
  long blocks = (n / 4 + 255) / 256;
  return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

extern "C" int carel_adam_step(const carel_adam_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a || !a->param || !a->grad || !a->exp_avg || !a->exp_avg_sq || a->n <= 0) return set_error(CAREL_ERR_ARG, "carel_adam_step: bad arguments");
  if (a->step < 1) return set_error(CAREL_ERR_ARG, "carel_adam_step: step must be >= 1");
  if (((uintptr_t)a->param | (uintptr_t)a->grad | (uintptr_t)a->exp_avg | (uintptr_t)a->exp_avg_sq) & 15)
    return set_error(CAREL_ERR_ARG, "carel_adam_step: buffers must be 16-byte aligned");
  AdamArgs k;
  k.p = (float*)a->param; k.g = (const float*)a->grad; k.m = (float*)a->exp_avg; k.v = (float*)a->exp_avg_sq;
  k.shadow = (bf16_t*)a->shadow_bf16; k.n = a->n;
  const double bc1 = 1.0 - pow((double)a->beta1, (double)a->step), bc2 = 1.0 - pow((double)a->beta2, (double)a->step);
  k.lr_over_bc1 = (float)((double)a->lr / bc1); k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  k.b1 = a->beta1; k.b2 = a->beta2; k.eps = a->eps; k.grad_scale = a->grad_scale == 0.f ? 1.f : a->grad_scale;
  k.skip_lo = a->skip_lo; k.skip_hi = a->skip_hi; k.skip_flag = (const float*)a->skip_flag;
  k.skip_count = (const float*)a->skip_count; k.lr = a->lr; k.step = (float)a->step;
  k.grad_scale_dev = (const float*)a->grad_scale_dev;
  k.decay = (float)(1.0 - (double)a->lr * (double)a->weight_decay);
  k.seg = (const long*)a->decay_segments; k.nseg = a->decay_segments ? a->n_decay_segments : 0;
  if (a->weight_decay != 0.f && !a->decay_segments) return set_error(CAREL_ERR_ARG, "carel_adam_step: weight_decay needs decay_segments");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(k.n, g_adam_grid_cap)), dim3(256), 0, stream, k);
  if (a->skip_count && a->skip_flag)       // after the update: one more frozen step on the range's record if this one was frozen
    hipLaunchKernelGGL(adam_skip_bump_kernel, dim3(1), dim3(1), 0, stream, (const float*)a->skip_flag, (float*)a->skip_count);
  return check_launch("adam_kernel");
}

extern "C" int carel_rmsprop_step(void* param, const void* grad, void* square_avg, int64_t n, float lr, float alpha, float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!param || !grad || !square_avg || n <= 0) return set_error(CAREL_ERR_ARG, "carel_rmsprop_step: bad arguments");
  long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (float*)param, (const float*)grad, (float*)square_avg, (long)n,
                     lr, alpha, eps);
  return check_launch("rmsprop_kernel");
}

extern "C" int carel_cast_f32_to_bf16(const void* src, void* dst, int64_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst || n <= 0) return set_error(CAREL_ERR_ARG, "carel_cast_f32_to_bf16: bad arguments");
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return set_error(CAREL_ERR_ARG, "carel_cast_f32_to_bf16: misaligned buffers");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, stream, (const float*)src, (bf16_t*)dst, (long)n);
  return check_launch("cast_bf16_kernel");
}
