// Micro-benchmark (measurement only): what does one s_barrier of a 512-thread workgroup cost inside a loop?
//   mode 0: bare loop of barriers          mode 1: s_setprio 1 / 0 around every second barrier (the ping-pong GEMM's pattern)
//   mode 2: as 1, waves 4..7 one barrier behind waves 0..3 (ping-pong groups)
//   mode 3: as 2 with ~200 VGPRs live (occupancy as the GEMM's)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/barrier tools/ubench/barrier.hip ; run: tools/ubench/barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int MODE>
__global__ __launch_bounds__(512) void bar(int iters, float* sink) {
  const int wave = threadIdx.x >> 6;
  float keep[MODE == 3 ? 160 : 1];
  if (MODE == 3) for (int i = 0; i < 160; ++i) keep[i] = (float)(threadIdx.x + i);
  if (MODE >= 2 && wave >= 4) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_barrier();
    if (MODE >= 1) __builtin_amdgcn_s_setprio(1);
    if (MODE == 3) for (int i = 0; i < 160; ++i) asm volatile("" : "+v"(keep[i]));
    if (MODE >= 1) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (MODE >= 2 && wave < 4) __builtin_amdgcn_s_barrier();
  if (MODE == 3) { float s = 0.f; for (int i = 0; i < 160; ++i) s += keep[i]; if (s == 12345.f) sink[0] = s; }
}
template <int MODE>
void run(const char* name, int wgs, float* sink) {
  const int iters = 2000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(bar<MODE>, dim3(wgs), dim3(512), 0, 0, iters, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(bar<MODE>, dim3(wgs), dim3(512), 0, 0, iters, sink);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-46s wgs %3d: %6.1f ns per barrier\n", name, wgs, ms * 1e6 / (5.0 * iters * 2));
}
int main() {
  float* sink; CHECK(hipMalloc(&sink, 64));
  for (int wgs : {256, 96}) {
    run<0>("bare barriers", wgs, sink);
    run<1>("+ s_setprio", wgs, sink);
    run<2>("+ two groups one barrier apart", wgs, sink);
    run<3>("+ 160 live registers", wgs, sink);
  }
  return 0;
}
