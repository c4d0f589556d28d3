// Sentence-embedding fine-tuning with the batch-semi-hard triplet loss: the arithmetic of chi_ec_sentence_transformer.py /
// en_ec_sentence_transformer.py (:22 SentenceTransformer, :78 losses.BatchSemiHardTripletLoss(model, margin), :84-87 .fit)
// that lives in the un-vendored third-party `sentence_transformers` (absent here: parity unpinned; the published algorithm
// is restated in oracle/carel_oracle_st.py):
//   mean pooling over the attended tokens of the encoder's last hidden states (Pooling, mode "mean") and its backward,
//   BatchSemiHardTripletLoss (Euclidean distance) forward + gradient in one workgroup,
//   the global gradient norm + clip coefficient of torch.nn.utils.clip_grad_norm_ (fit(): max_grad_norm = 1).
// All fp32.  The encoder itself is the same bf16 MFMA stack as the VAE path (carel_encoder_forward / _backward_layer).
#include "carel_hip_internal.h"

namespace carel {

// ---- mean pooling: out[b] = sum_{t < len_b} x[row0_b + t] / max(len_b, 1)   (prefix-form masks, HF right padding) -------
__global__ __launch_bounds__(256) void mean_pool_fwd_kernel(const float* __restrict__ x, const int* __restrict__ row0, const int* __restrict__ len,
                                                            float* __restrict__ out, int H) {
  const int b = blockIdx.x;
  const int n = len[b];
  const float* xb = x + (long)row0[b] * H;
  const float inv = 1.0f / (float)(n > 0 ? n : 1);
  for (int k = threadIdx.x; k < H; k += blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < n; ++t) s += xb[(long)t * H + k];
    out[(long)b * H + k] = s * inv;
  }
}
// dx[row] = g[b] / len_b for the rows of sample b, 0 for every other row (padding / filler rows): one thread per 4 columns
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const float* __restrict__ g, const int* __restrict__ row_sample, const int* __restrict__ len,
                                                            float* __restrict__ dx, long rows, int H) {
  const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (e >= rows * H) return;
  const long r = e / H;
  const int k = (int)(e - r * H);
  const int b = row_sample[r];
  float4 o = {0.f, 0.f, 0.f, 0.f};
  if (b >= 0) {
    const float inv = 1.0f / (float)(len[b] > 0 ? len[b] : 1);
    const float4 gv = *(const float4*)(g + (long)b * H + k);
    o = float4{gv.x * inv, gv.y * inv, gv.z * inv, gv.w * inv};
  }
  *(float4*)(dx + e) = o;
}

// ---- BatchSemiHardTripletLoss ---------------------------------------------------------------------------------------
// D[i][j] = Euclidean distance (0 where the squared distance is <= 0, with zero gradient there).  For every anchor a and
// positive p (same label, p != a): the negative distance is the SMALLEST D[a][n] over negatives n with D[a][n] > D[a][p]
// ("semi-hard"), or, when there is none, the LARGEST D[a][n] over all negatives (0 = D[a][a] when the batch has no negative
// for a); loss = sum max(D[a][p] - Dneg + margin, 0) / #positive pairs.  One workgroup; B <= 64 rows in LDS.
constexpr int TRIP_MAXB = 64;
__global__ __launch_bounds__(256) void triplet_semihard_kernel(const float* __restrict__ emb, const int* __restrict__ labels, int B, int H, float margin,
                                                               float* __restrict__ loss_out, float* __restrict__ demb) {
  __shared__ float D[TRIP_MAXB][TRIP_MAXB + 1];
  __shared__ float W[TRIP_MAXB][TRIP_MAXB + 1];     // d loss / d D[i][j] (both orientations accumulated: W[a][p] and W[a][n])
  __shared__ int lab[TRIP_MAXB];
  __shared__ float red[16];
  const int t = threadIdx.x;
  if (t < B) lab[t] = labels[t];
  for (int e = t; e < B * B; e += blockDim.x) {
    const int i = e / B, j = e - i * B;
    const float* a = emb + (long)i * H;
    const float* b = emb + (long)j * H;
    float dot = 0.f, na = 0.f, nb = 0.f;
    for (int k = 0; k < H; ++k) { dot = fmaf(a[k], b[k], dot); na = fmaf(a[k], a[k], na); nb = fmaf(b[k], b[k], nb); }
    const float d2 = na - 2.0f * dot + nb;            // square_norm[j] - 2 dot + square_norm[i], as the library orders it
    D[i][j] = (i == j || d2 <= 0.f) ? 0.f : sqrtf(d2);
    W[i][j] = 0.f;
  }
  __syncthreads();
  float part = 0.f, npos = 0.f;
  for (int e = t; e < B * B; e += blockDim.x) {
    const int a = e / B, p = e - a * B;
    if (a == p || lab[a] != lab[p]) continue;
    npos += 1.f;
    const float dap = D[a][p];
    int n_out = -1, n_in = -1;
    float d_out = 0.f, d_in = 0.f;
    for (int n = 0; n < B; ++n) {
      if (lab[n] == lab[a]) continue;
      const float dan = D[a][n];
      if (dan > dap && (n_out < 0 || dan < d_out)) { n_out = n; d_out = dan; }
      if (n_in < 0 || dan > d_in) { n_in = n; d_in = dan; }
    }
    const int nsel = n_out >= 0 ? n_out : n_in;       // -1: no negative at all -> distance 0, no gradient
    const float dneg = n_out >= 0 ? d_out : (n_in >= 0 ? d_in : 0.f);
    const float l = dap - dneg + margin;
    if (l > 0.f) {
      part += l;
      atomicAdd(&W[a][p], 1.0f);
      if (nsel >= 0) atomicAdd(&W[a][nsel], -1.0f);
    }
  }
  part = block_sum(part, red);
  npos = block_sum(npos, red + 8);
  const float inv = npos > 0.f ? 1.0f / npos : 0.f;    // (0 / 0 in the library when the batch has no positive pair: reported as 0 here)
  if (t == 0) loss_out[0] = part * inv;
  __syncthreads();
  if (!demb) return;
  // d D[i][j] / d e_i = (e_i - e_j) / D[i][j]  (0 where D = 0); every D[i][j] with weight W[i][j] moves BOTH e_i and e_j
  for (int e = t; e < B * H; e += blockDim.x) {
    const int i = e / H, k = e - i * H;
    const float ei = emb[(long)i * H + k];
    float g = 0.f;
    for (int j = 0; j < B; ++j) {
      const float w = W[i][j] + W[j][i];
      if (w != 0.f && D[i][j] > 0.f) g = fmaf(w / D[i][j], ei - emb[(long)j * H + k], g);
    }
    demb[e] = g * inv;
  }
}

// ---- global gradient norm and clip coefficient (torch.nn.utils.clip_grad_norm_(params, max_norm)) -----------------------
__global__ __launch_bounds__(256) void sqsum_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
  __shared__ float red[16];
  float s = 0.f;
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 4 <= n; i += stride) {
    const float4 v = *(const float4*)(g + i);
    s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
  }
  if (i < n && i + 4 > n) for (long j = i; j < n; ++j) s = fmaf(g[j], g[j], s);
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ part, int nparts, float max_norm, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s);
    const float coef = max_norm / (norm + 1e-6f);
    out[0] = norm;
    out[1] = coef < 1.0f ? coef : 1.0f;                // clip_coef_clamped
  }
}

// models.Normalize (all-mpnet-base-v2's third module): y = x / max(||x||_2, 1e-12) per row (torch.nn.functional.normalize), and its
// backward dx = (g - y (y . g)) / max(||x||, 1e-12).  One workgroup per row.
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ norm, int hidden) {
  __shared__ float red[8];
  const long r = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < hidden; c += 256) { const float v = x[r * hidden + c]; s += v * v; }
  s = block_sum(s, red);
  const float n = fmaxf(sqrtf(s), 1e-12f);
  if (threadIdx.x == 0) norm[r] = n;
  for (int c = threadIdx.x; c < hidden; c += 256) y[r * hidden + c] = x[r * hidden + c] / n;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ norm,
                                                         float* __restrict__ dx, int hidden) {
  __shared__ float red[8];
  const long r = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < hidden; c += 256) s += g[r * hidden + c] * y[r * hidden + c];
  s = block_sum(s, red);
  const float n = norm[r];
  for (int c = threadIdx.x; c < hidden; c += 256) dx[r * hidden + c] = (g[r * hidden + c] - y[r * hidden + c] * s) / n;
}

}  // namespace carel

using namespace carel;

extern "C" int carel_mean_pool_fwd(const void* x, const void* row0, const void* len, int32_t batch, int32_t hidden, void* out, void* stream) {
  if (!x || !row0 || !len || !out || batch < 1 || hidden < 1) return set_error(CAREL_ERR_ARG, "carel_mean_pool_fwd: bad arguments");
  hipLaunchKernelGGL(mean_pool_fwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const int*)row0, (const int*)len, (float*)out, hidden);
  return check_launch("mean_pool_fwd_kernel");
}
extern "C" int carel_mean_pool_bwd(const void* g, const void* row_sample, const void* len, int64_t rows, int32_t hidden, void* dx, void* stream) {
  if (!g || !row_sample || !len || !dx || rows < 1 || hidden < 4 || (hidden & 3)) return set_error(CAREL_ERR_ARG, "carel_mean_pool_bwd: bad arguments");
  const long q = rows * hidden / 4;
  hipLaunchKernelGGL(mean_pool_bwd_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)g, (const int*)row_sample,
                     (const int*)len, (float*)dx, (long)rows, hidden);
  return check_launch("mean_pool_bwd_kernel");
}
extern "C" int carel_l2_normalize_fwd(const void* x, int32_t rows, int32_t hidden, void* y, void* norm, void* stream) {
  if (!x || !y || !norm || rows < 1 || hidden < 1) return set_error(CAREL_ERR_ARG, "carel_l2_normalize_fwd: bad arguments");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, (float*)norm, hidden);
  return check_launch("l2norm_fwd_kernel");
}
extern "C" int carel_l2_normalize_bwd(const void* g, const void* y, const void* norm, int32_t rows, int32_t hidden, void* dx, void* stream) {
  if (!g || !y || !norm || !dx || rows < 1 || hidden < 1) return set_error(CAREL_ERR_ARG, "carel_l2_normalize_bwd: bad arguments");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const float*)g, (const float*)y, (const float*)norm, (float*)dx, hidden);
  return check_launch("l2norm_bwd_kernel");
}
extern "C" int carel_triplet_semihard(const void* emb, const void* labels, int32_t batch, int32_t hidden, float margin, void* loss_out, void* demb,
                                      void* stream) {
  if (!emb || !labels || !loss_out) return set_error(CAREL_ERR_ARG, "carel_triplet_semihard: null pointer");
  if (batch < 1 || batch > TRIP_MAXB || hidden < 1) return set_error(CAREL_ERR_SHAPE, "carel_triplet_semihard: batch must be in 1..%d", TRIP_MAXB);
  hipLaunchKernelGGL(triplet_semihard_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)emb, (const int*)labels, batch, hidden, margin,
                     (float*)loss_out, (float*)demb);
  return check_launch("triplet_semihard_kernel");
}
extern "C" int carel_grad_norm_clip(const void* grad, int64_t n, float max_norm, void* scratch, void* out2, void* stream) {
  if (!grad || !scratch || !out2 || n < 1) return set_error(CAREL_ERR_ARG, "carel_grad_norm_clip: bad arguments");
  if ((uintptr_t)grad & 15) return set_error(CAREL_ERR_ARG, "carel_grad_norm_clip: grad must be 16-byte aligned");
  const int blocks = 1024;                                       // scratch: 1024 floats
  hipLaunchKernelGGL(sqsum_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)grad, (long)n, (float*)scratch);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, blocks, max_norm, (float*)out2);
  return check_launch("clip_coef_kernel");
}
