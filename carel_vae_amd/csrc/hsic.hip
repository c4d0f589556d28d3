// Stand-alone HSIC operator (ablation head of drl_classifier_ec_hsic.py:529-547, :214) and its backward.
#include "carel_hip_internal.h"
#include "hsic_device.h"

namespace carel {

struct HsicArgs { const float* x; const float* y; long ldx, ldy; HsicCfg c; float* out; const float* grad; float* gx; float* gy; };

__global__ __launch_bounds__(1024) void hsic_kernel(HsicArgs a, int backward) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;
  const int m = a.c.m, d = a.c.d, xs = a.c.xs, mp = (m + 3) & ~3;
  float* nx = red + 64; float* ny = nx + mp; float* rk = ny + mp; float* rl = rk + mp;
  float* X = rl + mp; float* Y = X + m * xs;
  for (int e = threadIdx.x; e < m * d; e += blockDim.x) {
    const int i = e / d, k = e - i * d;
    X[i * xs + k] = a.x[(long)i * a.ldx + k]; Y[i * xs + k] = a.y[(long)i * a.ldy + k];
  }
  __syncthreads();
  const float h = hsic_forward_block(a.c, X, Y, nx, ny, rk, rl, red);
  if (!backward) { if (threadIdx.x == 0) a.out[0] = h; return; }
  const float tk = red[60], tl = red[61], gs = a.grad ? a.grad[0] : 1.f;
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    float gx[64], gy[64];
    hsic_backward_row(a.c, X, Y, nx, ny, rk, rl, tk, tl, i, gs, gx, gy);
    for (int k = 0; k < d; ++k) { a.gx[(long)i * d + k] = gx[k]; a.gy[(long)i * d + k] = gy[k]; }
  }
}

}  // namespace carel

using namespace carel;

static int hsic_launch(const carel_hsic_args* a, int backward, hipStream_t stream, const char* who) {
  if (!a || !a->x || !a->y) return set_error(CAREL_ERR_ARG, "%s: null sample pointer", who);
  if (a->m < 2 || a->d < 1 || a->d > 64) return set_error(CAREL_ERR_SHAPE, "%s: need m >= 2 and 1 <= d <= 64", who);
  if (!(a->s_x > 0.f) || !(a->s_y > 0.f)) return set_error(CAREL_ERR_ARG, "%s: kernel widths must be positive", who);
  if (!backward && !a->hsic_out) return set_error(CAREL_ERR_ARG, "%s: null output", who);
  if (backward && (!a->gx || !a->gy)) return set_error(CAREL_ERR_ARG, "%s: null gradient output", who);
  HsicArgs k;
  k.x = (const float*)a->x; k.y = (const float*)a->y; k.ldx = a->ldx; k.ldy = a->ldy;
  k.c.m = a->m; k.c.d = a->d; k.c.xs = a->d | 1; k.c.inv_sx = 1.0f / a->s_x; k.c.inv_sy = 1.0f / a->s_y;
  k.out = (float*)a->hsic_out; k.grad = (const float*)a->grad_hsic; k.gx = (float*)a->gx; k.gy = (float*)a->gy;
  const int mp = (a->m + 3) & ~3;
  const size_t lds = sizeof(float) * (64 + 4 * (size_t)mp + 2 * (size_t)a->m * k.c.xs);
  if (lds > 160 * 1024) return set_error(CAREL_ERR_SHAPE, "%s: m*d too large for one LDS", who);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)hsic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(hsic_kernel, dim3(1), dim3(1024), lds, stream, k, backward);
  return check_launch("hsic_kernel");
}

extern "C" int carel_hsic_fwd(const carel_hsic_args* a, void* stream) { return hsic_launch(a, 0, (hipStream_t)stream, "carel_hsic_fwd"); }
extern "C" int carel_hsic_bwd(const carel_hsic_args* a, void* stream) { return hsic_launch(a, 1, (hipStream_t)stream, "carel_hsic_bwd"); }
