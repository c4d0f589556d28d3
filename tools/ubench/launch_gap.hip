// Micro-benchmark (measurement only): the cost of a kernel boundary in one stream.
//   (a) N launches of an empty kernel (1 workgroup) back to back: time per launch = dispatch + completion of a dependent kernel
//   (b) the same with 256 workgroups of 512 threads and 128 KiB of LDS each (the GEMM's footprint), each spinning ~20 us:
//       per-launch time minus the spin = what a boundary costs between two chip-filling kernels
//   (c) as (b) but every workgroup records s_memrealtime at entry and exit: gap = first entry of launch i+1 - last exit of launch i
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/launch_gap tools/ubench/launch_gap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void empty_k() {}
__global__ __launch_bounds__(512) void spin_k(unsigned long long* rec, int launch, int spin) {
  extern __shared__ char smem[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) smem[0] = 1;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
  __syncthreads();
  if (threadIdx.x == 0 && rec) { rec[((long)launch * gridDim.x + blockIdx.x) * 2] = t0; rec[((long)launch * gridDim.x + blockIdx.x) * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}

int main() {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms;
  const int N = 2000;
  for (int r = 0; r < 2; ++r) {
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, 0);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  }
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("(a) empty kernel, %d back-to-back launches: %.2f us per launch\n", N, ms * 1e3 / N);
  CHECK(hipFuncSetAttribute((const void*)spin_k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  const int L = 200, G = 256;
  unsigned long long* rec; CHECK(hipMalloc(&rec, (size_t)L * G * 16));
  for (int spin : {0, 12}) {
    for (int r = 0; r < 2; ++r) {
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < L; ++i) hipLaunchKernelGGL(spin_k, dim3(G), dim3(512), 131072, 0, rec, i, spin);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    }
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)L * G * 2);
    CHECK(hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost));
    double gap = 0, life = 0, span = 0; std::vector<double> gaps;
    for (int i = 0; i < L; ++i) {
      unsigned long long s0 = ~0ull, s1 = 0, e_first = ~0ull, e_last = 0;
      for (int b = 0; b < G; ++b) { s0 = std::min(s0, h[((size_t)i * G + b) * 2]); s1 = std::max(s1, h[((size_t)i * G + b) * 2]); e_last = std::max(e_last, h[((size_t)i * G + b) * 2 + 1]); life += (double)(h[((size_t)i * G + b) * 2 + 1] - h[((size_t)i * G + b) * 2]); }
      span += (double)(e_last - s0);
      if (i + 1 < L) { unsigned long long n0 = ~0ull; for (int b = 0; b < G; ++b) n0 = std::min(n0, h[((size_t)(i + 1) * G + b) * 2]); gaps.push_back((double)n0 - (double)e_last); }
    }
    std::sort(gaps.begin(), gaps.end());
    printf("(b/c) 256 workgroups x 512 threads x 128 KiB LDS, spin %d: %.2f us per launch (events); workgroup life %.2f us, first entry -> last exit %.2f us, "
           "gap last exit -> next first entry: median %.2f us (min %.2f, max %.2f)\n", spin, ms * 1e3 / L, life / (L * G) * 0.01, span / L * 0.01, gaps[gaps.size() / 2] * 0.01, gaps.front() * 0.01, gaps.back() * 0.01);
  }
  return 0;
}
