// Stand-alone RBF-MMD operator: the HIP body of MMDStatistic.__call__ / pdist
// (drl_classifier_ec_mmd_final_mul.py:547-569, :580-589) and of its autograd backward.
// One workgroup of 1024 threads; both samples live in LDS (padded odd row stride), the Gram matrix is
// never written unless ret_matrix asks for it; block sums use wave shuffles + an LDS stage.
#include "carel_hip_internal.h"
#include "mmd_device.h"

namespace carel {

struct MmdKernelArgs {
  const float* s1; const float* s2; long ld1, ld2;
  MmdCfg cfg;
  float* mmd_out; float* kernels_out;
  const float* grad_mmd; float* g1; float* g2;
};

__device__ __forceinline__ void mmd_load(const MmdKernelArgs& a, float* Z) {
  const MmdCfg& c = a.cfg;
  const int n = c.n1 + c.n2;
  for (int e = threadIdx.x; e < n * c.d; e += blockDim.x) {
    const int i = e / c.d, k = e - i * c.d;
    Z[i * c.zs + k] = (i < c.n1) ? a.s1[(long)i * a.ld1 + k] : a.s2[(long)(i - c.n1) * a.ld2 + k];
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void mmd_fwd_kernel(MmdKernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;            // 64 floats
  float* nrm = red + 64;
  const int n = a.cfg.n1 + a.cfg.n2;
  float* Z = nrm + ((n + 3) & ~3);
  mmd_load(a, Z);
  const float mmd = mmd_forward_block(a.cfg, Z, nrm, red, a.kernels_out);
  if (threadIdx.x == 0) a.mmd_out[0] = mmd;
}

template <int G>
__global__ __launch_bounds__(1024) void mmd_bwd_kernel(MmdKernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;
  float* nrm = red + 64;
  const MmdCfg& c = a.cfg;
  const int n = c.n1 + c.n2;
  float* Z = nrm + ((n + 3) & ~3);
  mmd_load(a, Z);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < c.d; ++k) s = fmaf(Z[i * c.zs + k], Z[i * c.zs + k], s);
    nrm[i] = s;
  }
  __syncthreads();
  const float gs = a.grad_mmd ? a.grad_mmd[0] : 1.0f;
  const int rows_per_pass = blockDim.x / G;
  const int grp = threadIdx.x % G;
  for (int base = 0; base < n; base += rows_per_pass) {     // trip count is uniform across the block
    const int i = base + threadIdx.x / G;
    float g[64];
    const bool live = i < n;
    if (live) mmd_backward_row(c, Z, nrm, i, gs, g, grp, G);
    else for (int k = 0; k < c.d; ++k) g[k] = 0.f;
    for (int k = 0; k < c.d; ++k) {
      float v = g[k];
#pragma unroll
      for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
      if (live && grp == 0) {
        if (i < c.n1) a.g1[(long)i * c.d + k] = v; else a.g2[(long)(i - c.n1) * c.d + k] = v;
      }
    }
  }
}

// ---- pdist (ref :580-589, the live norm = 2 branch): dist[i][j] = sqrt(eps + |n1_i + n2_j - 2 s1_i . s2_j|) ----------------
// One wave per row i; lane = feature k (d <= 64), so the dot product is one wave reduction per partner j.  A utility of the
// drop-in surface, not on the training path (the fused tail / MMD kernels never materialise distances).
struct PdistArgs { const float* s1; const float* s2; long ld1, ld2; int n1, n2, d; float eps; float* out; const float* gout; float* g1; float* g2; };

__global__ __launch_bounds__(64) void pdist_fwd_kernel(PdistArgs a) {
  const int i = blockIdx.x, k = threadIdx.x;
  const float x = k < a.d ? a.s1[(long)i * a.ld1 + k] : 0.f;
  const float nx = wave_sum(x * x);
  for (int j = 0; j < a.n2; ++j) {
    const float y = k < a.d ? a.s2[(long)j * a.ld2 + k] : 0.f;
    const float ny = wave_sum(y * y), dot = wave_sum(x * y);
    if (k == 0) a.out[(long)i * a.n2 + j] = sqrtf(a.eps + fabsf(nx + ny - 2.0f * dot));
  }
}
// SIDE 0: g1[i][k] = sum_j G_ij sign(d2_ij) (s1_ik - s2_jk) / dist_ij ; SIDE 1: g2[j][k] = -sum_i (same summand)
template <int SIDE>
__global__ __launch_bounds__(64) void pdist_bwd_kernel(PdistArgs a) {
  const int r = blockIdx.x, k = threadIdx.x;
  const float* mine = SIDE == 0 ? a.s1 + (long)r * a.ld1 : a.s2 + (long)r * a.ld2;
  const float x = k < a.d ? mine[k] : 0.f;
  const float nx = wave_sum(x * x);
  const int nother = SIDE == 0 ? a.n2 : a.n1;
  float g = 0.f;
  for (int o = 0; o < nother; ++o) {
    const float* other = SIDE == 0 ? a.s2 + (long)o * a.ld2 : a.s1 + (long)o * a.ld1;
    const float y = k < a.d ? other[k] : 0.f;
    const float ny = wave_sum(y * y), dot = wave_sum(x * y);
    const float d2 = nx + ny - 2.0f * dot;
    const float dist = sqrtf(a.eps + fabsf(d2));
    const float up = SIDE == 0 ? a.gout[(long)r * a.n2 + o] : a.gout[(long)o * a.n2 + r];
    const float sg = d2 > 0.f ? 1.f : (d2 < 0.f ? -1.f : 0.f);
    g = fmaf(up * sg / dist, x - y, g);        // d|d2|/d mine = sign * 2 (mine - other); d sqrt = 1 / (2 dist)
  }
  if (k < a.d) (SIDE == 0 ? a.g1 : a.g2)[(long)r * a.d + k] = g;
}

}  // namespace carel

using namespace carel;

static int pdist_prepare(const carel_pdist_args* a, PdistArgs* k, const char* who) {
  if (!a || !a->s1 || !a->s2) return set_error(CAREL_ERR_ARG, "%s: null sample pointer", who);
  if (a->n1 < 1 || a->n2 < 1) return set_error(CAREL_ERR_SHAPE, "%s: empty sample", who);
  if (a->d < 1 || a->d > 64) return set_error(CAREL_ERR_SHAPE, "%s: d must be in 1..64 (got %d)", who, a->d);
  k->s1 = (const float*)a->s1; k->s2 = (const float*)a->s2; k->ld1 = a->ld1; k->ld2 = a->ld2;
  k->n1 = a->n1; k->n2 = a->n2; k->d = a->d; k->eps = a->eps;
  k->out = (float*)a->dist_out; k->gout = (const float*)a->grad_dist; k->g1 = (float*)a->g1; k->g2 = (float*)a->g2;
  return CAREL_OK;
}
extern "C" int carel_pdist_fwd(const carel_pdist_args* a, void* stream) {
  PdistArgs k;
  int rc = pdist_prepare(a, &k, "carel_pdist_fwd");
  if (rc) return rc;
  if (!k.out) return set_error(CAREL_ERR_ARG, "carel_pdist_fwd: null dist_out");
  hipLaunchKernelGGL(pdist_fwd_kernel, dim3(k.n1), dim3(64), 0, (hipStream_t)stream, k);
  return check_launch("pdist_fwd_kernel");
}
extern "C" int carel_pdist_bwd(const carel_pdist_args* a, void* stream) {
  PdistArgs k;
  int rc = pdist_prepare(a, &k, "carel_pdist_bwd");
  if (rc) return rc;
  if (!k.gout || !k.g1 || !k.g2) return set_error(CAREL_ERR_ARG, "carel_pdist_bwd: null gradient pointer");
  hipLaunchKernelGGL(pdist_bwd_kernel<0>, dim3(k.n1), dim3(64), 0, (hipStream_t)stream, k);
  hipLaunchKernelGGL(pdist_bwd_kernel<1>, dim3(k.n2), dim3(64), 0, (hipStream_t)stream, k);
  return check_launch("pdist_bwd_kernel");
}

static int mmd_prepare(const carel_mmd_args* a, MmdKernelArgs* k, size_t* lds, const char* who) {
  if (!a || !a->s1 || !a->s2) return set_error(CAREL_ERR_ARG, "%s: null sample pointer", who);
  if (a->n1 < 2 || a->n2 < 2) return set_error(CAREL_ERR_SHAPE, "%s: n1,n2 must be >= 2 (the reference divides by n(n-1))", who);
  if (a->d < 1 || a->d > 64) return set_error(CAREL_ERR_SHAPE, "%s: d must be in 1..64 (got %d)", who, a->d);
  if (a->n_alphas < 1 || a->n_alphas > 8) return set_error(CAREL_ERR_ARG, "%s: n_alphas must be in 1..8", who);
  const int n = a->n1 + a->n2;
  const int zs = a->d | 1;
  *lds = (size_t)(64 + ((n + 3) & ~3) + (size_t)n * zs) * sizeof(float);
  if (*lds > 160 * 1024) return set_error(CAREL_ERR_SHAPE, "%s: (n1+n2)*d too large for one LDS (%zu bytes)", who, *lds);
  k->s1 = (const float*)a->s1; k->s2 = (const float*)a->s2; k->ld1 = a->ld1; k->ld2 = a->ld2;
  k->cfg.n1 = a->n1; k->cfg.n2 = a->n2; k->cfg.d = a->d; k->cfg.zs = zs; k->cfg.n_alphas = a->n_alphas;
  for (int i = 0; i < 8; ++i) k->cfg.alphas[i] = i < a->n_alphas ? a->alphas[i] : 0.f;
  k->cfg.eps = a->eps;
  k->mmd_out = (float*)a->mmd_out; k->kernels_out = (float*)a->kernels_out;
  k->grad_mmd = (const float*)a->grad_mmd; k->g1 = (float*)a->g1; k->g2 = (float*)a->g2;
  return CAREL_OK;
}

template <typename K>
static int set_lds(K kern, size_t lds, const char* who) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
  }
  return CAREL_OK;
}

extern "C" int carel_rbf_mmd_fwd(const carel_mmd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MmdKernelArgs k; size_t lds;
  int rc = mmd_prepare(a, &k, &lds, "carel_rbf_mmd_fwd");
  if (rc) return rc;
  if (!k.mmd_out) return set_error(CAREL_ERR_ARG, "carel_rbf_mmd_fwd: null mmd_out");
  if ((rc = set_lds(mmd_fwd_kernel, lds, "carel_rbf_mmd_fwd"))) return rc;
  hipLaunchKernelGGL(mmd_fwd_kernel, dim3(1), dim3(1024), lds, stream, k);
  return check_launch("mmd_fwd_kernel");
}

extern "C" int carel_rbf_mmd_bwd(const carel_mmd_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MmdKernelArgs k; size_t lds;
  int rc = mmd_prepare(a, &k, &lds, "carel_rbf_mmd_bwd");
  if (rc) return rc;
  if (!k.g1 || !k.g2) return set_error(CAREL_ERR_ARG, "carel_rbf_mmd_bwd: null gradient output");
  const int n = k.cfg.n1 + k.cfg.n2;
  if (n * 8 <= 1024) {
    if ((rc = set_lds(mmd_bwd_kernel<8>, lds, "carel_rbf_mmd_bwd"))) return rc;
    hipLaunchKernelGGL(mmd_bwd_kernel<8>, dim3(1), dim3(1024), lds, stream, k);
  } else if (n * 2 <= 1024) {
    if ((rc = set_lds(mmd_bwd_kernel<2>, lds, "carel_rbf_mmd_bwd"))) return rc;
    hipLaunchKernelGGL(mmd_bwd_kernel<2>, dim3(1), dim3(1024), lds, stream, k);
  } else {
    if ((rc = set_lds(mmd_bwd_kernel<1>, lds, "carel_rbf_mmd_bwd"))) return rc;
    hipLaunchKernelGGL(mmd_bwd_kernel<1>, dim3(1), dim3(1024), lds, stream, k);
  }
  return check_launch("mmd_bwd_kernel");
}
