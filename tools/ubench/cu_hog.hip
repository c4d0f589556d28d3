// Measurement helper (round 4): `n` workgroups that do nothing but stay resident for `usec` microseconds -- a stand-in for the persistent
// channel workgroups of a collective kernel running beside the training step (what does a step made of ONE-round GEMMs,
// 256 tiles on 256 CUs, lose when a few CUs are taken?).  Every workgroup leaves when the wall clock says so: the grid always drains.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libcu_hog.so tools/ubench/cu_hog.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void cu_hog_kernel(long ticks, int lds_words, unsigned* sink) {
  extern __shared__ unsigned lds[];
  const long t0 = (long)wall_clock64();
  unsigned acc = 0;
  while ((long)wall_clock64() - t0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    if (lds_words) acc += lds[threadIdx.x % lds_words];
  }
  if (acc == 0xdeadbeefu) sink[0] = acc;
}
extern "C" int cu_hog_launch(int n, double usec, int lds_bytes, void* sink, void* stream) {
  int dev = 0, rate_khz = 100000;
  if (hipGetDevice(&dev) != hipSuccess) return 1;
  if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
  const long ticks = (long)(usec * 1e-3 * (double)rate_khz);
  if (usec > 50000.0) return 2;                                   // never more than 50 ms
  hipLaunchKernelGGL(cu_hog_kernel, dim3(n), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, ticks, lds_bytes / 4, (unsigned*)sink);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
