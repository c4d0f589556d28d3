// Micro-benchmark (measurement only): the GEMM epilogue's store pattern against whole-row stores.
// A [8192][3072] bf16 output is written tile by tile (256 x 192 per workgroup of 512 threads, 512 tiles on 256 workgroups x 2):
//   mode 0: as gemm_pp's epilogue -- a wave instruction = 16 rows x 64 contiguous bytes (lane -> row l & 15, 16-B piece l >> 4)
//   mode 1: whole rows -- a wave instruction = 8 rows x 128 contiguous bytes
//   mode 2: a wave instruction = 4 rows x 256 contiguous bytes (the tile row is 384 B: a full and a half-populated instruction)
//   mode 3: a wave instruction = 2 rows x 384 B (the tile's full row: 24 lanes per row; 48 of 64 lanes active)
// NOUT = 1 or 2 output arrays (the GELU epilogue writes two).  Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/store_pattern ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int M = 8192, N = 3072, TM = 256, TN = 192;

template <int MODE, int NOUT>
__global__ __launch_bounds__(512) void store_k(unsigned short* o0, unsigned short* o1, int tiles_n, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const long base = (long)tm * TM * N + (long)tn * TN;
    if (MODE == 0) {          // wave (wr, wc): rows wr*64.., cols wc*96..; 4 row blocks x 3 column groups of 32
      const int wr = wave & 3, wc = wave >> 2;
      for (int b = 0; b < 4; ++b)
        for (int q = 0; q < 3; ++q) {
          const long off = base + (long)(wr * 64 + b * 16 + (lane & 15)) * N + wc * 96 + q * 32 + (lane >> 4) * 8;
          *(uint4*)(o0 + off) = v;
          if (NOUT == 2) *(uint4*)(o1 + off) = v;
        }
    } else {
      constexpr int SEG = MODE == 1 ? 128 : MODE == 2 ? 256 : 384;      // contiguous bytes per row per instruction
      constexpr int LPR = SEG / 16;                                     // lanes per row
      constexpr int RPI = MODE == 3 ? 2 : 64 / LPR;                     // rows per instruction
      constexpr int CPR = (384 + SEG - 1) / SEG;                        // instructions per row span (the last one partly populated: 384 = 1.5 x 256)
      // the wave owns 32 rows of the tile (8 waves x 32 = 256)
      const bool act = lane < RPI * LPR;
      for (int r = 0; r < 32; r += RPI)
        for (int c = 0; c < CPR; ++c) {
          const long off = base + (long)(wave * 32 + r + lane / LPR) * N + (c * SEG + (lane % LPR) * 16) / 2;
          if (act && c * SEG + (lane % LPR) * 16 < 384) { *(uint4*)(o0 + off) = v; if (NOUT == 2) *(uint4*)(o1 + off) = v; }
        }
    }
  }
}

template <int MODE, int NOUT>
void run(const char* name, unsigned short* o0, unsigned short* o1) {
  const int tiles_n = N / TN, ntiles = (M / TM) * tiles_n;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((store_k<MODE, NOUT>), dim3(256), dim3(512), 0, 0, o0, o1, tiles_n, ntiles);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((store_k<MODE, NOUT>), dim3(256), dim3(512), 0, 0, o0, o1, tiles_n, ntiles);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 20.0 * NOUT * (double)M * N * 2;
  printf("%-34s outputs %d: %6.1f us per launch  %5.2f TB/s\n", name, NOUT, ms * 1e3 / 20, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  unsigned short *o0, *o1;
  CHECK(hipMalloc(&o0, (size_t)M * N * 2)); CHECK(hipMalloc(&o1, (size_t)M * N * 2));
  run<0, 1>("16 rows x 64 B (gemm_pp epilogue)", o0, o1); run<0, 2>("16 rows x 64 B (gemm_pp epilogue)", o0, o1);
  run<1, 1>("8 rows x 128 B", o0, o1); run<1, 2>("8 rows x 128 B", o0, o1);
  run<2, 1>("4 rows x 256 B", o0, o1); run<2, 2>("4 rows x 256 B", o0, o1);
  run<3, 1>("2 rows x 384 B (48 lanes)", o0, o1); run<3, 2>("2 rows x 384 B (48 lanes)", o0, o1);
  return 0;
}
