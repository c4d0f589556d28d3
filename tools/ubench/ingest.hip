// Micro-benchmark (measurement only, not part of the library): how fast can ONE CU pull L2-resident tile rows
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction = 8 rows x 128 B)
//   mode 1: global_load_dwordx4 into registers (consumed by an xor)
//   mode 2: global_load_dwordx4 into registers, then ds_write_b128 into LDS
//   mode 3: waves 0-3 as mode 0 and waves 4-7 as mode 2 (both paths at once)
// The source is a [rows][pitch] bf16 matrix of `foot` bytes that every workgroup sweeps (so it is served by L2 / MALL).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/ingest tools/ubench/ingest.hip ; run: tools/ubench/ingest
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define LDSP __attribute__((address_space(3)))
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void ingest(const char* src, long pitch, int rows, int iters, unsigned* sink, int nw_active) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r8 = lane >> 3, c = lane & 7;
  // a wave instruction = 8 consecutive rows x 128 B at column block `cb`; the waves of a WG take consecutive 8-row pieces
  uint4 acc = {0, 0, 0, 0};
  const int pieces = rows / 8;
  const int cbs = (int)(pitch / 128);
  int piece = (blockIdx.x * 37 + wave) % pieces, cb = blockIdx.x % cbs;
  char* my_lds = smem + wave * (DEPTH * 1024);
  const bool dma = MODE == 0 || (MODE == 3 && wave < 4);
  if (wave >= nw_active) iters = 0;            // only the first nw_active waves move data (how many issuing waves does a CU need?)
  for (int it = 0; it < iters; ++it) {
    if (dma) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const char* g = src + (long)(piece * 8 + r8) * pitch + cb * 128 + c * 16;
        __builtin_amdgcn_global_load_lds((const void*)g, (LDSP void*)(my_lds + d * 1024), 16, 0, 0);
        piece += 8; if (piece >= pieces) { piece -= pieces; cb = cb + 1 == cbs ? 0 : cb + 1; }
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
    } else {
      uint4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const char* g = src + (long)(piece * 8 + r8) * pitch + cb * 128 + c * 16;
        v[d] = *(const uint4*)g;
        piece += 8; if (piece >= pieces) { piece -= pieces; cb = cb + 1 == cbs ? 0 : cb + 1; }
      }
      if (MODE == 1) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
      } else {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) *(uint4*)(my_lds + d * 1024 + lane * 16) = v[d];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 1) { const uint4 t = *(uint4*)(smem + threadIdx.x * 16); acc.x ^= t.x ^ t.y ^ t.z ^ t.w; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

template <int MODE, int DEPTH>
void run(const char* name, const char* src, long pitch, int rows, unsigned* sink, int wgs, int nw = 8) {
  const int iters = 400;
  const size_t lds = 8 * DEPTH * 1024;
  CHECK(hipFuncSetAttribute((const void*)ingest<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((ingest<MODE, DEPTH>), dim3(wgs), dim3(512), lds, 0, src, pitch, rows, iters, sink, nw);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((ingest<MODE, DEPTH>), dim3(wgs), dim3(512), lds, 0, src, pitch, rows, iters, sink, nw);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 5.0 * wgs * (double)nw * iters * DEPTH * 1024.0;
  printf("%-28s waves %d depth %2d wgs %4d: %7.1f GB/s per CU   %6.2f TB/s chip\n", name, nw, DEPTH, wgs, bytes / (ms * 1e-3) / 1e9 / (wgs < 256 ? wgs : 256), bytes / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const long foot = argc > 1 ? atol(argv[1]) : (2L << 20);      // bytes swept by every WG
  const long pitch = argc > 2 ? atol(argv[2]) : 1536;
  const int rows = (int)(foot / pitch) / 64 * 64;
  char* src; unsigned* sink;
  CHECK(hipMalloc(&src, (size_t)rows * pitch)); CHECK(hipMemset(src, 1, (size_t)rows * pitch));
  CHECK(hipMalloc(&sink, 4));
  printf("footprint %.1f MB, pitch %ld B, %d rows\n", rows * pitch / 1e6, pitch, rows);
  for (int nw : {1, 2, 4, 6, 8}) {          // how the LDS-DMA rate of a CU grows with the number of waves that issue it
    run<0, 4>("LDS-DMA", src, pitch, rows, sink, 256, nw);
    run<0, 8>("LDS-DMA", src, pitch, rows, sink, 256, nw);
  }
  for (int wgs : {256, 512}) {
    run<0, 4>("LDS-DMA", src, pitch, rows, sink, wgs);
    run<0, 8>("LDS-DMA", src, pitch, rows, sink, wgs);
    run<0, 16>("LDS-DMA", src, pitch, rows, sink, wgs);
    run<1, 4>("global_load -> VGPR", src, pitch, rows, sink, wgs);
    run<1, 8>("global_load -> VGPR", src, pitch, rows, sink, wgs);
    run<1, 16>("global_load -> VGPR", src, pitch, rows, sink, wgs);
    run<2, 4>("global_load -> VGPR -> LDS", src, pitch, rows, sink, wgs);
    run<2, 8>("global_load -> VGPR -> LDS", src, pitch, rows, sink, wgs);
    run<3, 8>("4 waves DMA + 4 waves VGPR", src, pitch, rows, sink, wgs);
  }
  return 0;
}
