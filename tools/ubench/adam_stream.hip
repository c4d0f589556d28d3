// Micro-benchmark (measurement only, round 4): how fast can an Adam-shaped pass stream?  Per element: p, g, m, v read (16 B), p, m, v + a bf16
// shadow written (14 B) = 30 B; 103.5 M elements = 3.1 GB.  Variants of the access pattern around the product kernel's (csrc/adam.hip):
//   mode 0: grid-stride, one float4 per array per thread per iteration (the product kernel's pattern), GRID blocks of 256
//   mode 1: two float4 per array in flight per thread (i and i + stride)
//   mode 2: each block streams ONE contiguous chunk (n / gridDim) instead of striding over the whole range
//   mode 3: mode 0 with non-temporal loads of g (read once) and non-temporal stores of everything
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/adam_stream tools/ubench/adam_stream.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void upd(float4& p, const float4 g, float4& m, float4& v) {
  float* pp = &p.x; const float* gg = &g.x; float* mm = &m.x; float* vv = &v.x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mm[e] = mm[e] + 0.1f * (gg[e] - mm[e]);
    vv[e] = vv[e] * 0.999f + 0.001f * gg[e] * gg[e];
    pp[e] = pp[e] - 1e-5f * (mm[e] / (sqrtf(vv[e]) * 1.01f + 1e-8f));
  }
}
__device__ __forceinline__ uint2 pk(const float4 p) {
  typedef __attribute__((ext_vector_type(2))) float f2; typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  const b2 a = __builtin_convertvector(f2{p.x, p.y}, b2), b = __builtin_convertvector(f2{p.z, p.w}, b2);
  return uint2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}

template <int MODE>
__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                              unsigned short* __restrict__ sh, long n) {
  if (MODE == 2) {
    const long per = ((n / 4 + gridDim.x - 1) / gridDim.x) * 4;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (long i = lo + threadIdx.x * 4; i < hi; i += 1024) {
      float4 P = *(float4*)(p + i); const float4 G = *(const float4*)(g + i); float4 M = *(float4*)(m + i), V = *(float4*)(v + i);
      upd(P, G, M, V);
      *(float4*)(p + i) = P; *(float4*)(m + i) = M; *(float4*)(v + i) = V; *(uint2*)(sh + i) = pk(P);
    }
    return;
  }
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * 1024;
  if (MODE == 1) {
    for (; i + stride < n; i += 2 * stride) {
      float4 P0 = *(float4*)(p + i), P1 = *(float4*)(p + i + stride);
      const float4 G0 = *(const float4*)(g + i), G1 = *(const float4*)(g + i + stride);
      float4 M0 = *(float4*)(m + i), M1 = *(float4*)(m + i + stride), V0 = *(float4*)(v + i), V1 = *(float4*)(v + i + stride);
      upd(P0, G0, M0, V0); upd(P1, G1, M1, V1);
      *(float4*)(p + i) = P0; *(float4*)(m + i) = M0; *(float4*)(v + i) = V0; *(uint2*)(sh + i) = pk(P0);
      *(float4*)(p + i + stride) = P1; *(float4*)(m + i + stride) = M1; *(float4*)(v + i + stride) = V1; *(uint2*)(sh + i + stride) = pk(P1);
    }
  }
  for (; i < n; i += stride) {
    float4 P = *(float4*)(p + i);
    float4 G;
    if (MODE == 3) { const float* gp = g + i; G = float4{__builtin_nontemporal_load(gp), __builtin_nontemporal_load(gp + 1), __builtin_nontemporal_load(gp + 2), __builtin_nontemporal_load(gp + 3)}; }
    else G = *(const float4*)(g + i);
    float4 M = *(float4*)(m + i), V = *(float4*)(v + i);
    upd(P, G, M, V);
    if (MODE == 3) {
      typedef __attribute__((ext_vector_type(4))) float f4;
      __builtin_nontemporal_store(f4{P.x, P.y, P.z, P.w}, (f4*)(p + i)); __builtin_nontemporal_store(f4{M.x, M.y, M.z, M.w}, (f4*)(m + i));
      __builtin_nontemporal_store(f4{V.x, V.y, V.z, V.w}, (f4*)(v + i));
      *(uint2*)(sh + i) = pk(P);
    } else {
      *(float4*)(p + i) = P; *(float4*)(m + i) = M; *(float4*)(v + i) = V; *(uint2*)(sh + i) = pk(P);
    }
  }
}

int main() {
  const long n = 103500032;
  float *p, *g, *m, *v; unsigned short* sh;
  CHECK(hipMalloc(&p, n * 4)); CHECK(hipMalloc(&g, n * 4)); CHECK(hipMalloc(&m, n * 4)); CHECK(hipMalloc(&v, n * 4)); CHECK(hipMalloc(&sh, n * 2));
  CHECK(hipMemset(p, 0, n * 4)); CHECK(hipMemset(g, 0, n * 4)); CHECK(hipMemset(m, 0, n * 4)); CHECK(hipMemset(v, 0, n * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grids[] = {4096, 32768, 50538, 65536, 101075};
  for (int mode = 0; mode < 4; ++mode)
    for (int gi = 0; gi < 5; ++gi) {
      const int grid = grids[gi];
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(adam_k<0>, dim3(grid), dim3(256), 0, 0, p, g, m, v, sh, n);
        if (mode == 1) hipLaunchKernelGGL(adam_k<1>, dim3(grid), dim3(256), 0, 0, p, g, m, v, sh, n);
        if (mode == 2) hipLaunchKernelGGL(adam_k<2>, dim3(grid), dim3(256), 0, 0, p, g, m, v, sh, n);
        if (mode == 3) hipLaunchKernelGGL(adam_k<3>, dim3(grid), dim3(256), 0, 0, p, g, m, v, sh, n);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("mode %d grid %5d: %7.1f us  %.2f TB/s\n", mode, grid, best * 1e3f, 30.0 * n / (best * 1e-3) / 1e12);
    }
  return 0;
}
