// Column sums of per-block partials (LayerNorm backward: dgamma / dbeta / bias gradient; embeddings), shared by ln.hip and by the
// slab reduction of gemm.hip, which runs the LayerNorm partials of a layer in the same launch as the weight-gradient slabs.
#pragma once
#include "carel_common.h"

namespace carel {

// out[c] (+)= sum_p partials[p][c]   (c < n, p < nparts): 16 columns x 16 row-lanes per block, fixed order
// Column sums of partials [nparts][n]: PR_COLS columns x 8 part-lanes per block (128-byte segments, four loads in
// flight per thread), fixed summation order.  Returns the sum in the threads of part-lane 0; c = column.
constexpr int PR_COLS = 32;
__device__ __forceinline__ float partial_colsum16(const float* __restrict__ partials, int n, int nparts, float* lds, int& c, int blk) {
  const int cl = threadIdx.x & (PR_COLS - 1), rl = threadIdx.x / PR_COLS;      // rl in [0, 8)
  c = blk * PR_COLS + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < n) {
    int p = rl;
    for (; p + 24 < nparts; p += 32) {
      s0 += partials[(long)p * n + c]; s1 += partials[(long)(p + 8) * n + c];
      s2 += partials[(long)(p + 16) * n + c]; s3 += partials[(long)(p + 24) * n + c];
    }
    for (; p < nparts; p += 8) s0 += partials[(long)p * n + c];
  }
  lds[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  float t = 0.f;
  if (rl == 0) {
#pragma unroll
    for (int r = 0; r < 256 / PR_COLS; ++r) t += lds[r * PR_COLS + cl];
  }
  return t;
}

// column c goes to outs.p[c / seg][c % seg] (null pointers are skipped)
struct SegOuts { float* p[4]; };

}  // namespace carel
