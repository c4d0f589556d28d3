// Hardware-layout self test: dumps what the MFMA / transposed-LDS-read / LDS-DMA helpers of
// carel_common.h actually produce on exact small-integer data, so tests/test_gpu_layouts.py can check
// every layout assumption the GEMM and attention kernels are built on (and say WHICH one broke).
#include "carel_hip_internal.h"

namespace carel {

// input (bf16 elements)                      output (f32 elements)
//  [0,512)        A16 [16][32]                [0,256)       C16 = A16*B16            [16][16]
//  [512,1024)     B16 [32][16]                [256,1280)    C32 = A32*B32            [32][32]
//  [1024,1536)    A32 [32][16]                [1280,1536)   tr dump: lane*4+e
//  [1536,2048)    B32 [16][32]                [2048,2560)   LDS-DMA dump (512 bf16 of LDS)
//  [2048,3072)    A2  [32][32]                [4096,5120)   Y = A2 * X, X = A32*B32  [32][32]
//  [4096,12288)   TA_row [128 m][64 k]        [8192 + c*16384 ...) C tile of combo c = 0 NT,1 NN,2 TN,3 TT
//  [12288,20480)  TB_row [128 n][64 k]
//  [20480,28672)  TA_col [64 k][128 m]
//  [28672,36864)  TB_col [64 k][128 n]
//  [36864,40960)  TR [32][128]
__global__ __launch_bounds__(64) void selftest_wave_kernel(const bf16_t* __restrict__ in, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[32 * 128 + 1024];
  const int l = threadIdx.x;
  // ---- 0: mfma 16x16x32: A[row=l&15][k=8(l>>4)+j], B[k=8(l>>4)+j][col=l&15]; D col=l&15,row=(l>>4)*4+r
  {
    s16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (short)in[(l & 15) * 32 + 8 * (l >> 4) + j];
      b[j] = (short)in[512 + (8 * (l >> 4) + j) * 16 + (l & 15)];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = mfma16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
    for (int r = 0; r < 4; ++r) out[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
  }
  // ---- 1: mfma 32x32x16: A[row=l&31][k=8(l>>5)+j], B[k=8(l>>5)+j][col=l&31]
  f32x16 x;
  {
    s16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (short)in[1024 + (l & 31) * 16 + 8 * (l >> 5) + j];
      b[j] = (short)in[1536 + (8 * (l >> 5) + j) * 32 + (l & 31)];
    }
    for (int r = 0; r < 16; ++r) x[r] = 0.f;
    x = mfma32(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), x);
    for (int r = 0; r < 16; ++r) out[256 + acc32_row(r, l) * 32 + (l & 31)] = x[r];
  }
  // ---- 2: ds_read_b64_tr_b16 raw semantics
  for (int e = l; e < 32 * 128; e += 64) lds[e] = in[36864 + e];
  __syncthreads();
  {
    const int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    const bf16_t* addr = lds + (4 * g + q) * 128 + 16 * g + 4 * p;
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((CAREL_LDS s16x4*)addr);
    for (int e = 0; e < 4; ++e) out[1280 + l * 4 + e] = bf2f((bf16_t)t[e]);
  }
  __syncthreads();
  // ---- 3: LDS-DMA: lane l fetches global chunk (l ^ 5) of 16 B; LDS destination must be lane-linear
  {
    bf16_t* dst = lds + 32 * 128;     // 1 KiB region
    const bf16_t* src = in + 4096 + ((l ^ 5) * 8);
    __builtin_amdgcn_global_load_lds(src, (CAREL_LDS void*)dst, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = l; e < 512; e += 64) out[2048 + e] = bf2f(dst[e]);
  }
  // ---- 4: accumulator tile as the B operand of the next MFMA:  Y = A2 * X
  {
    f32x16 y;
    for (int r = 0; r < 16; ++r) y[r] = 0.f;
    const int h = l >> 5;
    for (int s = 0; s < 2; ++s) {
      s16x8 a, b;
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
        a[j] = (short)in[2048 + (l & 31) * 32 + k];
        b[j] = (short)f2bf(x[8 * s + j]);
      }
      y = mfma32(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), y);
    }
    for (int r = 0; r < 16; ++r) out[4096 + acc32_row(r, l) * 32 + (l & 31)] = y[r];
  }
}

// ---- 5: one 128x128x64 block product through the real staging + fragment helpers, 4 operand combos
template <bool AT, bool BT>
__global__ __launch_bounds__(256) void selftest_tile_kernel(const bf16_t* __restrict__ in, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) char smem[32768];
  char* ta = smem;
  char* tb = smem + 16384;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  if (AT) stage_col_image(in + 20480, 128, 0, 0, ta); else stage_row_image<128>(in + 4096, 64, 0, 0, ta);
  if (BT) stage_col_image(in + 28672, 128, 0, 0, tb); else stage_row_image<128>(in + 12288, 64, 0, 0, tb);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 64; ks += 32) {
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i] = AT ? frag16_col(ta, wr * 64 + i * 16, ks) : frag16_row(ta, wr * 64 + i * 16, ks);
      fb[i] = BT ? frag16_col(tb, wc * 64 + i * 16, ks) : frag16_row(tb, wc * 64 + i * 16, ks);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
  }
  const int combo = (AT ? 2 : 0) + (BT ? 1 : 0);
  float* o = out + 8192 + combo * 16384;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 4; ++r)
        o[(wr * 64 + i * 16 + (lane & 15)) * 128 + wc * 64 + j * 16 + (lane >> 4) * 4 + r] = acc[i][j][r];
}

}  // namespace carel

using namespace carel;

extern "C" int carel_selftest_layouts(const void* in_bf16, void* out_f32, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!in_bf16 || !out_f32) return set_error(CAREL_ERR_ARG, "carel_selftest_layouts: null buffer");
  const bf16_t* in = (const bf16_t*)in_bf16;
  float* out = (float*)out_f32;
  hipLaunchKernelGGL(selftest_wave_kernel, dim3(1), dim3(64), 0, stream, in, out);
  hipLaunchKernelGGL((selftest_tile_kernel<false, false>), dim3(1), dim3(256), 0, stream, in, out);
  hipLaunchKernelGGL((selftest_tile_kernel<false, true>), dim3(1), dim3(256), 0, stream, in, out);
  hipLaunchKernelGGL((selftest_tile_kernel<true, false>), dim3(1), dim3(256), 0, stream, in, out);
  hipLaunchKernelGGL((selftest_tile_kernel<true, true>), dim3(1), dim3(256), 0, stream, in, out);
  return check_launch("selftest kernels");
}
