// Micro-benchmark (measurement only): where do the workgroups of a one-workgroup-per-CU kernel (128 KiB of LDS, 512 threads) land,
// and do they all run at once?  Every workgroup records XCC_ID, HW_ID (SE / SH / CU) and its start / end time; the host prints,
// per grid size, the number of distinct CUs used, the largest number of workgroups that shared one CU, and the span of the launch
// against one workgroup's duration (1.0 = everything ran concurrently, 2.0 = some CU ran two workgroups back to back).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/placement tools/ubench/placement.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };

__global__ __launch_bounds__(512) void probe(Rec* out, int spin) {
  extern __shared__ char smem[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) smem[0] = 1;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);          // ~1.7 us per iteration at 2.4 GHz
  __syncthreads();
  if (threadIdx.x == 0) {
    Rec r;
    r.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xf;          // HW_REG_XCC_ID
    r.hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));                // HW_REG_HW_ID
    r.t0 = t0; r.t1 = wall_clock64();
    out[blockIdx.x] = r;
  }
}

int main() {
  Rec* d; CHECK(hipMalloc(&d, 4096 * sizeof(Rec)));
  CHECK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  std::vector<Rec> h(4096);
  for (int grid : {64, 96, 108, 128, 144, 160, 192, 216, 240, 252, 256, 288, 512}) {
    hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 131072, 0, d, 12);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 131072, 0, d, 12);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d, grid * sizeof(Rec), hipMemcpyDeviceToHost));
    std::map<unsigned, int> per_cu; std::map<unsigned, int> per_xcc;
    unsigned long long tmin = ~0ull, tmax = 0, dsum = 0;
    for (int i = 0; i < grid; ++i) {
      const unsigned cu = (h[i].hwid >> 8) & 0xf, sh = (h[i].hwid >> 12) & 1, se = (h[i].hwid >> 13) & 7;
      per_cu[(h[i].xcc << 16) | (se << 8) | (sh << 4) | cu]++; per_xcc[h[i].xcc]++;
      tmin = std::min(tmin, h[i].t0); tmax = std::max(tmax, h[i].t1); dsum += h[i].t1 - h[i].t0;
    }
    int worst = 0; for (auto& kv : per_cu) worst = std::max(worst, kv.second);
    int xmin = 1 << 30, xmax = 0; for (auto& kv : per_xcc) { xmin = std::min(xmin, kv.second); xmax = std::max(xmax, kv.second); }
    printf("grid %4d: %3zu distinct CUs, most workgroups on one CU %d, per XCD %d..%d, launch span / workgroup duration %.2f\n", grid, per_cu.size(), worst,
           xmin, xmax, (double)(tmax - tmin) / ((double)dsum / grid));
  }
  return 0;
}
