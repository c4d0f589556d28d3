// Row-vector kernels shared by the VAE tails (tail.hip, en_tail.hip): fp32 linear layers whose input width is the encoder
// hidden size (pooler, latent heads), their weight / data gradients, and small helpers.
#pragma once
#include "carel_hip_internal.h"

namespace carel {

constexpr int TH = 768;
constexpr int TNV = TH / 256;

// ------------------------------------------------------------------------------------------
// out[b][n] = act( sum_k in[b*in_stride + k] * W_n[k] + bias_n )   one wave per output column n,
// W_n = w[n / seg] + (n % seg) * K ; bias likewise (lets the 4 latent heads share one launch)
// ------------------------------------------------------------------------------------------
struct PtrSet4 { const float* w[4]; const float* b[4]; };

template <int ACT>   // 0 none, 1 tanh
__global__ __launch_bounds__(256) void rowvec_linear_kernel(const float* __restrict__ in, long in_stride, const int* __restrict__ row_idx,
                                                            int B, int N, int seg, PtrSet4 ps, float* __restrict__ out, long out_stride) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  // blockIdx.y: contiguous group of samples (more waves in flight than one wave per column)
  const int bg = (B + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * bg, b1 = min(B, b0 + bg);
  const float* w = ps.w[n / seg] + (long)(n % seg) * TH;
  const float bias = ps.b[n / seg] ? ps.b[n / seg][n % seg] : 0.f;
  float4 wv[TNV];
#pragma unroll
  for (int i = 0; i < TNV; ++i) wv[i] = *(const float4*)(w + (i * 64 + lane) * 4);
  for (int b = b0; b < b1; b += 4) {        // four samples per trip: independent loads and reductions
    float s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int bb = min(b + u, b1 - 1);
      const float* x = row_idx ? in + (long)row_idx[bb] * TH : in + (long)bb * in_stride;
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < TNV; ++i) {
        const float4 xv = *(const float4*)(x + (i * 64 + lane) * 4);
        t += (xv.x * wv[i].x + xv.y * wv[i].y) + (xv.z * wv[i].z + xv.w * wv[i].w);
      }
      s[u] = t;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += __shfl_xor(s[u], off, 64);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v = s[u] + bias;
      if (ACT == 1) v = tanhf(v);
      if (lane == 0 && b + u < b1) out[(long)(b + u) * out_stride + n] = v;
    }
  }
}

// dW_n[k] = sum_b dY[b][n] * X[b*x_stride + k] ; db_n = sum_b dY[b][n]   (one wave per n)
struct OutSet4 { float* w[4]; float* b[4]; };
static __global__ __launch_bounds__(256) void rowvec_wgrad_kernel(const float* __restrict__ dY, long dy_stride, const float* __restrict__ X,
                                                           long x_stride, const int* __restrict__ row_idx, int B, int N, int seg, OutSet4 os) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float4 acc[TNV];
#pragma unroll
  for (int i = 0; i < TNV; ++i) acc[i] = float4{0.f, 0.f, 0.f, 0.f};
  float sb = 0.f;
  constexpr int U = 8;                      // samples in flight per trip (round 4: 4 -> 8; every trip is one round trip to memory: 64 samples were 16 of them)
  for (int b = 0; b < B; b += U) {          // accumulation order stays b = 0, 1, 2, ...
    float g[U]; float4 xv[U][TNV];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int bb = min(b + u, B - 1);
      g[u] = (b + u < B) ? dY[(long)bb * dy_stride + n] : 0.f;
      const float* x = row_idx ? X + (long)row_idx[bb] * TH : X + (long)bb * x_stride;
#pragma unroll
      for (int i = 0; i < TNV; ++i) xv[u][i] = *(const float4*)(x + (i * 64 + lane) * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sb += g[u];
#pragma unroll
      for (int i = 0; i < TNV; ++i) {
        acc[i].x = fmaf(g[u], xv[u][i].x, acc[i].x); acc[i].y = fmaf(g[u], xv[u][i].y, acc[i].y);
        acc[i].z = fmaf(g[u], xv[u][i].z, acc[i].z); acc[i].w = fmaf(g[u], xv[u][i].w, acc[i].w);
      }
    }
  }
  float* w = os.w[n / seg] + (long)(n % seg) * TH;
#pragma unroll
  for (int i = 0; i < TNV; ++i) *(float4*)(w + (i * 64 + lane) * 4) = acc[i];
  if (lane == 0 && os.b[n / seg]) os.b[n / seg][n % seg] = sb;
}

// part[c][b][k] = sum_{n in chunk c} dY[b][n] * W_n[k]  (one wave per (sample b, chunk of 64 outputs n));
// MODE 1: dY is first multiplied by (1 - y^2) of the tanh output y (pooler) and the product is also written
// to dpre (for the pooler wgrad).  The chunks are summed by sum_parts_kernel (fixed order).
constexpr int DG_CHUNK = 32;          // (round 4: 64 -> 32 outputs per wave, eight weight rows in flight: four trips to memory per wave instead of 16 -- pooler data gradient 28 -> ~9 us)
template <int MODE>
__global__ __launch_bounds__(256) void rowvec_dgrad_kernel(const float* __restrict__ dY, long dy_stride, int B, int N, int seg,
                                                           PtrSet4 ps, const float* __restrict__ y, float* __restrict__ dpre,
                                                           float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float4 acc[TNV];
#pragma unroll
  for (int i = 0; i < TNV; ++i) acc[i] = float4{0.f, 0.f, 0.f, 0.f};
  const int n0 = blockIdx.y * DG_CHUNK, n1 = min(N, n0 + DG_CHUNK);
  constexpr int U = 8;
  for (int nb = n0; nb < n1; nb += U) {     // eight weight rows in flight; accumulation order stays n = n0, n0+1, ...
    float g[U]; float4 wv[U][TNV];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = min(nb + u, n1 - 1);
      float gg = (nb + u < n1) ? dY[(long)b * dy_stride + n] : 0.f;
      if (MODE == 1) {
        const float yy = y[(long)b * N + n];
        gg *= (1.0f - yy * yy);
        if (lane == 0 && nb + u < n1) dpre[(long)b * N + n] = gg;
      }
      g[u] = gg;
      const float* w = ps.w[n / seg] + (long)(n % seg) * TH;
#pragma unroll
      for (int i = 0; i < TNV; ++i) wv[u][i] = *(const float4*)(w + (i * 64 + lane) * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < TNV; ++i) {
        acc[i].x = fmaf(g[u], wv[u][i].x, acc[i].x); acc[i].y = fmaf(g[u], wv[u][i].y, acc[i].y);
        acc[i].z = fmaf(g[u], wv[u][i].z, acc[i].z); acc[i].w = fmaf(g[u], wv[u][i].w, acc[i].w);
      }
    }
  }
  float* o = part + ((long)blockIdx.y * B + b) * TH;
#pragma unroll
  for (int i = 0; i < TNV; ++i) *(float4*)(o + (i * 64 + lane) * 4) = acc[i];
}

static __global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ part, float* __restrict__ out, long n, int nparts) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = *(const float4*)(part + i);
  for (int p = 1; p < nparts; ++p) {
    const float4 b = *(const float4*)(part + (long)p * n + i);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  *(float4*)(out + i) = a;
}

// x[i] *= *scale (device scalar: the grad_output of loss.backward(), no host sync needed)
static __global__ void scale_inplace_kernel(float* __restrict__ x, long n, const float* __restrict__ scale) {
  const float s = scale[0];
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) x[i] *= s;
}
// scatter the CLS-row gradients into the [T, 768] gradient of the last encoder output (zeroed first)
static __global__ __launch_bounds__(256) void scatter_cls_kernel(const float* __restrict__ dcls, int B, int S, const int* __restrict__ row_idx,
                                                          float* __restrict__ dx) {
  const int b = blockIdx.x;
  const long row = row_idx ? (long)row_idx[b] : (long)b * S;
  for (int k = threadIdx.x; k < TH; k += 256) dx[row * TH + k] = dcls[(long)b * TH + k];
}


}  // namespace carel
