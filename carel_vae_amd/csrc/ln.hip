// Row-wise kernels of the encoder: embeddings + LayerNorm (HF BertEmbeddings / RobertaEmbeddings),
// sub-layer LayerNorm forward/backward (BertSelfOutput.LayerNorm, BertOutput.LayerNorm) with the
// sub-layer dropout's backward fused in, embedding backward (scatter-add), and column sums (bias grads).
// All are HBM-bound: one wave per 768-wide row, 16-byte accesses, 1 KiB per wave instruction.
// Reached from drl_classifier_ec_mmd_final_mul.py:202-206 (forward) and :841 (backward).
#include "carel_hip_internal.h"
#include "ln_device.h"
#include "reduce_device.h"

namespace carel {

// ------------------------------------------------------------------------------------------
// LayerNorm forward:  x = LN(h)  -> x_f32 (residual stream), x_bf16 (next GEMM operand), stats
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ h, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, long rows,
                                                     float* __restrict__ x_f32, bf16_t* __restrict__ x_bf16,
                                                     float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  Row X = load_row(h + row * H, lane);
  const Row G = load_row(gamma, lane), Bt = load_row(beta, lane);
  float mean, rstd;
  ln_normalise(X, G, Bt, eps, mean, rstd);
  if (x_f32) store_row(x_f32 + row * H, lane, X);
  if (x_bf16) store_row_bf16(x_bf16 + row * H, lane, X);
  if (stats && lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
}

// LayerNorm forward behind a split-K out-projection / FFN2 (packed ECPE batches, ~1.8 k rows): the slab epilogue
// h = dropout(sum_z slabs[z] + bias) + residual -- slab_epilogue_kernel<EPI_BIAS_DROP_RESID>'s expression, additions in its order, the residual
// either stored f32 rows or LN(resid) recomputed from the previous LayerNorm's input and statistics like the GEMM epilogues do -- and the
// LayerNorm of h in ONE launch: h is written once (the backward pass and the next residual read it) and never read back here.
struct LnFwdSlabs {
  const float* slabs; int splits; long plane;
  const float* bias; const float* resid; const float* resid_stats; const float* resid_gamma; const float* resid_beta;
  Dropout drop; const int* row_map;
  float* h_out;
};
__global__ __launch_bounds__(256) void ln_fwd_slabs_kernel(LnFwdSlabs q, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, long rows,
                                                           float* __restrict__ x_f32, bf16_t* __restrict__ x_bf16, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  Row X = load_row(q.slabs + row * H, lane);
  Row Rr = load_row(q.resid + row * H, lane);
  for (int z = 1; z < q.splits; ++z) {
    const Row Z = load_row(q.slabs + (long)z * q.plane + row * H, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) { X.v[i].x += Z.v[i].x; X.v[i].y += Z.v[i].y; X.v[i].z += Z.v[i].z; X.v[i].w += Z.v[i].w; }
  }
  if (q.bias) {
    const Row Bs = load_row(q.bias, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) { X.v[i].x += Bs.v[i].x; X.v[i].y += Bs.v[i].y; X.v[i].z += Bs.v[i].z; X.v[i].w += Bs.v[i].w; }
  }
  if (q.resid_stats) {                 // the residual is LN(resid): the expression of ln_normalise / the GEMM epilogues (ln_apply)
    const float pm = q.resid_stats[row * 2], pr = q.resid_stats[row * 2 + 1];
    Row& A = Rr; const Row B = load_row(q.resid_gamma, lane), Cc = load_row(q.resid_beta, lane);
    ROW_FOREACH(a = ln_apply(a, pm, pr, b, c); (void)e)
  }
  const long drow = q.row_map ? (long)q.row_map[row] : row;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float dm[4];
    dropout_mult_n<4>(q.drop, (uint32_t)(drow * H + (i * 64 + lane) * 4), dm);
    X.v[i].x = X.v[i].x * dm[0] + Rr.v[i].x; X.v[i].y = X.v[i].y * dm[1] + Rr.v[i].y;
    X.v[i].z = X.v[i].z * dm[2] + Rr.v[i].z; X.v[i].w = X.v[i].w * dm[3] + Rr.v[i].w;
  }
  if (q.h_out) store_row(q.h_out + row * H, lane, X);
  const Row G = load_row(gamma, lane), Bt = load_row(beta, lane);
  float mean, rstd;
  ln_normalise(X, G, Bt, eps, mean, rstd);
  if (x_f32) store_row(x_f32 + row * H, lane, X);
  if (x_bf16) store_row_bf16(x_bf16 + row * H, lane, X);
  if (stats && lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
}

// ------------------------------------------------------------------------------------------
// Embeddings:  x0 = dropout(LN(word[ids] + pos[pid] + type[tt]))
// position ids: BERT arange(S); RoBERTa cumsum(ids != pad) * (ids != pad) + pad
// ------------------------------------------------------------------------------------------
struct EmbedArgs {
  const long* ids; const long* tt;
  const float* word; const float* pos; const float* type;
  const float* gamma; const float* beta;
  float eps; int S; long rows; int roberta; int pad_id; int vocab, max_pos, type_vocab;
  Dropout drop;
  const int* tok_row;     // packed: original row of output row t (-1 = filler); null = identity
};

__device__ __forceinline__ int position_id(const EmbedArgs& a, long row, int lane) {
  const int s = (int)(row % a.S);
  if (!a.roberta) return s;
  const long base = row - s;
  int cnt = 0;
  for (int c0 = 0; c0 <= s; c0 += 64) {
    const int j = c0 + lane;
    const bool nz = (j <= s) && (a.ids[base + j] != a.pad_id);
    cnt += __popcll(__ballot(nz));
  }
  const bool self_nz = a.ids[row] != a.pad_id;
  return (self_nz ? cnt : 0) + a.pad_id;
}

__device__ __forceinline__ Row embed_gather(const EmbedArgs& a, long row, int lane, int& pid, int& tid_) {
  long id = a.ids[row];
  id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);        // clamp: never read out of bounds
  pid = position_id(a, row, lane);
  pid = pid >= a.max_pos ? a.max_pos - 1 : pid;
  long t = a.tt ? a.tt[row] : 0;
  tid_ = (int)(t < 0 ? 0 : (t >= a.type_vocab ? a.type_vocab - 1 : t));
  Row A = load_row(a.word + id * H, lane);
  const Row B = load_row(a.pos + (long)pid * H, lane), Cc = load_row(a.type + (long)tid_ * H, lane);
  ROW_FOREACH(a = a + b + c; (void)e)
  return A;
}

__global__ __launch_bounds__(256) void embed_fwd_kernel(EmbedArgs a, float* __restrict__ x_f32, bf16_t* __restrict__ x_bf16,
                                                        float* __restrict__ stats, int2* __restrict__ row_keys) {
  const int lane = threadIdx.x & 63;
  const long orow = (long)blockIdx.x * 4 + (threadIdx.x >> 6);     // output row
  if (orow >= a.rows) return;
  const long row = a.tok_row ? (long)a.tok_row[orow] : orow;        // original row b*S + s
  if (row_keys && lane == 0 && row < 0) row_keys[orow] = int2{-1, -1};
  if (row < 0) {                                                    // filler row of a packed batch
    Row Z;
#pragma unroll
    for (int i = 0; i < NV; ++i) Z.v[i] = float4{0.f, 0.f, 0.f, 0.f};
    store_row(x_f32 + orow * H, lane, Z);
    store_row_bf16(x_bf16 + orow * H, lane, Z);
    if (lane == 0) { stats[orow * 2] = 0.f; stats[orow * 2 + 1] = 0.f; }
    return;
  }
  int pid, tid_;
  Row X = embed_gather(a, row, lane, pid, tid_);
  if (row_keys && lane == 0) {                                      // (token id, position id) of this row: what embed_sort_kernel sorts by
    long id = a.ids[row];
    id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
    row_keys[orow] = int2{(int)id, pid};
  }
  const Row G = load_row(a.gamma, lane), Bt = load_row(a.beta, lane);
  float mean, rstd;
  ln_normalise(X, G, Bt, a.eps, mean, rstd);
  if (a.drop.thresh) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const uint32_t e0 = (uint32_t)(row * H + (i * 64 + lane) * 4);
      float dm[4];
      dropout_mult_n<4>(a.drop, e0, dm);
      X.v[i].x *= dm[0]; X.v[i].y *= dm[1]; X.v[i].z *= dm[2]; X.v[i].w *= dm[3];
    }
  }
  store_row(x_f32 + orow * H, lane, X);
  store_row_bf16(x_bf16 + orow * H, lane, X);
  if (lane == 0) { stats[orow * 2] = mean; stats[orow * 2 + 1] = rstd; }
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward core on one row: dxhat = dy*g ; dh = rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat))
// On return DY holds dh, XH holds xhat*dy_in (the dgamma contribution), and dbeta contribution is dy_in
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void ln_bwd_row(Row& DY /*in: dy, out: dh*/, Row& XH /*in: h, out: dy*xhat*/, const Row& G,
                                           float mean, float rstd) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float* d = &DY.v[i].x; float* x = &XH.v[i].x; const float* g = &G.v[i].x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (x[e] - mean) * rstd;
      const float dxh = d[e] * g[e];
      s1 += dxh; s2 += dxh * xh;
      x[e] = xh;
    }
  }
  s1 = wave_sum(s1) * (1.0f / H);
  s2 = wave_sum(s2) * (1.0f / H);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float* d = &DY.v[i].x; float* x = &XH.v[i].x; const float* g = &G.v[i].x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dy = d[e], xh = x[e];
      d[e] = rstd * (dy * g[e] - s1 - xh * s2);
      x[e] = dy * xh;
    }
  }
}

constexpr int LNB_ROWS = 16;   // rows per block in the backward kernels (4 per wave)

// combine the 4 waves' column partials through LDS and write them: part[blk][slot][H]
__device__ __forceinline__ void write_partials(float* lds /*[4][H]*/, const Row& acc, float* __restrict__ dst, int lane, int wave) {
  __syncthreads();
  store_row(lds + wave * H, lane, acc);
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) dst[c] = (lds[c] + lds[H + c]) + (lds[2 * H + c] + lds[3 * H + c]);
}

// dy_out: grad w.r.t. the LN output (f32).  Outputs: dh (f32, grad w.r.t. the LN input = residual-path
// gradient), dyb (bf16, dh * dropout-mask of the sub-layer output = grad w.r.t. the GEMM result),
// partials[blk][3][H] = {dgamma, dbeta, dbias}.
// SLAB (round 4, packed ECPE batches): dy_out is not a stored array but the deferred epilogue of a split-K data-gradient GEMM --
// sum_z slabs[z][row] (+ resid[row]), added in the order slab_epilogue_kernel<CAREL_EPI_ADD_F32> uses -- so that the f32 rows never make a
// round trip through memory and one launch per sub-layer goes away (at ~1.8 k rows every launch is ~6-10 us of mostly latency).
struct LnSlabSrc { int splits; long plane; const float* resid; };
template <bool SLAB, int RPW = 4>        // RPW rows per wave, 4 * RPW rows per workgroup (ln_bwd_rpw: fewer at packed row counts)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy_out, const float* __restrict__ h,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     long rows, Dropout drop, const int* __restrict__ row_map,
                                                     float* __restrict__ dh, bf16_t* __restrict__ dyb,
                                                     float* __restrict__ partials, LnSlabSrc src) {
  __shared__ float lds[4 * H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Row G = load_row(gamma, lane);
  Row accG, accB, accBias;
#pragma unroll
  for (int i = 0; i < NV; ++i) accG.v[i] = accB.v[i] = accBias.v[i] = float4{0.f, 0.f, 0.f, 0.f};
  // all of this wave's rows are requested up front: with one row in flight per wave (2 waves per SIMD at 8192 rows) the kernel ran at the
  // latency of its load -> reduce -> store chain (17.4 us for 88 MB); same arithmetic, same summation order
  constexpr int WG_ROWS = 4 * RPW;
  Row DYr[RPW], XHr[RPW];
  float mr[RPW], rr[RPW];
  int dr[RPW];
  // (branch-free: rows past the end re-read the last row, a null row_map reads the stats array instead -- a divergent or uniform branch
  // around these loads makes the compiler drain vmcnt at every join, one row at a time again)
  const int* rm = row_map ? row_map : (const int*)stats;
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const long row = (long)blockIdx.x * WG_ROWS + wave + 4 * k;
    const long lrow = row < rows ? row : rows - 1;
    DYr[k] = load_row(dy_out + lrow * H, lane);
    if constexpr (SLAB) {
      for (int z = 1; z < src.splits; ++z) {
        const Row Z = load_row(dy_out + (long)z * src.plane + lrow * H, lane);
#pragma unroll
        for (int i = 0; i < NV; ++i) { DYr[k].v[i].x += Z.v[i].x; DYr[k].v[i].y += Z.v[i].y; DYr[k].v[i].z += Z.v[i].z; DYr[k].v[i].w += Z.v[i].w; }
      }
      if (src.resid) {
        const Row R = load_row(src.resid + lrow * H, lane);
#pragma unroll
        for (int i = 0; i < NV; ++i) { DYr[k].v[i].x += R.v[i].x; DYr[k].v[i].y += R.v[i].y; DYr[k].v[i].z += R.v[i].z; DYr[k].v[i].w += R.v[i].w; }
      }
    }
    XHr[k] = load_row(h + lrow * H, lane);
    mr[k] = stats[lrow * 2]; rr[k] = stats[lrow * 2 + 1];
    dr[k] = rm[lrow];                               // (a load issued after a row's stores would wait for their acknowledgement)
  }
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const long row = (long)blockIdx.x * WG_ROWS + wave + 4 * k;
    if (row >= rows) break;
    Row DY = DYr[k];
    Row XH = XHr[k];
    const float mean = mr[k], rstd = rr[k];
#pragma unroll
    for (int i = 0; i < NV; ++i) { accB.v[i].x += DY.v[i].x; accB.v[i].y += DY.v[i].y; accB.v[i].z += DY.v[i].z; accB.v[i].w += DY.v[i].w; }
    ln_bwd_row(DY, XH, G, mean, rstd);
#pragma unroll
    for (int i = 0; i < NV; ++i) { accG.v[i].x += XH.v[i].x; accG.v[i].y += XH.v[i].y; accG.v[i].z += XH.v[i].z; accG.v[i].w += XH.v[i].w; }
    if (dh) store_row(dh + row * H, lane, DY);
    if (drop.thresh) {
      const long drow = row_map ? (long)dr[k] : row;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const uint32_t e0 = (uint32_t)(drow * H + (i * 64 + lane) * 4);
        float dm[4];
        dropout_mult_n<4>(drop, e0, dm);
        DY.v[i].x *= dm[0]; DY.v[i].y *= dm[1]; DY.v[i].z *= dm[2]; DY.v[i].w *= dm[3];
      }
    }
    if (dyb) store_row_bf16(dyb + row * H, lane, DY);
#pragma unroll
    for (int i = 0; i < NV; ++i) { accBias.v[i].x += DY.v[i].x; accBias.v[i].y += DY.v[i].y; accBias.v[i].z += DY.v[i].z; accBias.v[i].w += DY.v[i].w; }
  }
  float* dst = partials + (long)blockIdx.x * 3 * H;
  write_partials(lds, accG, dst, lane, wave);
  write_partials(lds, accB, dst + H, lane, wave);
  write_partials(lds, accBias, dst + 2 * H, lane, wave);
}

// dpos[s][c] = sum_b rows[(b*S + s)][c]   (dense BERT positions; one thread per 4 columns, fixed order over the batch)
__global__ __launch_bounds__(256) void pos_reduce_kernel(const float* __restrict__ rows, int B, int S, float* __restrict__ dpos) {
  const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= (long)S * H) return;
  float4 acc = float4{0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < B; b += 4) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (b + u < B) ? *(const float4*)(rows + (long)(b + u) * S * H + e) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  *(float4*)(dpos + e) = acc;
}

// ------------------------------------------------------------------------------------------
// The table gradients WITHOUT atomics (round 4): d word_emb[v] = sum of the gradient rows of the tokens with id v, in row order.
// embed_sort_kernel (forward pass; the ids are known there, so the sort is off the backward pass's critical path): one workgroup per
// table (0 = token ids, 1 = position ids) sorts the keys id << 13 | row of up to 8192 rows in the LDS (bitonic network); filler rows
// and the padding up to the next power of two carry 0xFFFFFFFF and end up last.  embed_segsum_kernel (backward pass): one wave per
// (sorted position, 256-column chunk); the wave at the FIRST position of a run of equal ids adds the run's gradient rows in ascending row
// order (the low key bits) and writes the table row with plain stores -- a fixed order of additions: bit-reproducible, unlike the
// 6.3 M float atomics this replaces (92 us per step at T = 8192, and the library's last order-dependent sum).
// ------------------------------------------------------------------------------------------
constexpr int EMB_SORT_MAX = 8192, EMB_ROW_BITS = 13;
__global__ __launch_bounds__(1024) void embed_sort_kernel(const int2* __restrict__ row_keys, int rows, uint32_t* __restrict__ out) {
  __shared__ uint32_t k[EMB_SORT_MAX];
  const int which = (int)blockIdx.x;
  int n = 64;
  while (n < rows) n <<= 1;                                 // rows <= EMB_SORT_MAX (host-checked)
  for (int i = threadIdx.x; i < n; i += 1024) {
    uint32_t key = 0xFFFFFFFFu;
    if (i < rows) {
      const int2 v = row_keys[i];
      const int id = which ? v.y : v.x;
      if (id >= 0) key = ((uint32_t)id << EMB_ROW_BITS) | (uint32_t)i;
    }
    k[i] = key;
  }
  __syncthreads();
  for (int size = 2; size <= n; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (n >> 1); t += 1024) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint32_t a = k[lo], b = k[hi];
        if ((a > b) == up) { k[lo] = b; k[hi] = a; }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < EMB_SORT_MAX; i += 1024) out[which * EMB_SORT_MAX + i] = i < n ? k[i] : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void embed_segsum_kernel(const uint32_t* __restrict__ keys, const float* __restrict__ rows, float* __restrict__ table,
                                                           int table_rows) {
  const int lane = threadIdx.x & 63;
  const int w = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  const int pos = w / 3, chunk = w - pos * 3;
  if (pos >= EMB_SORT_MAX) return;
  const uint32_t key = keys[pos];
  if (key == 0xFFFFFFFFu) return;
  const uint32_t id = key >> EMB_ROW_BITS;
  if ((int)id >= table_rows) return;
  if (pos > 0 && (keys[pos - 1] >> EMB_ROW_BITS) == id) return;          // not the first position of its run
  const float* src = rows + chunk * 256 + lane * 4;
  float4 acc = float4{0.f, 0.f, 0.f, 0.f};
  for (int base = pos; base < EMB_SORT_MAX; base += 64) {
    // 64 keys of the run at a time: the row indices are then known up front and the row loads are independent of each other
    const uint32_t kj = base + lane < EMB_SORT_MAX ? keys[base + lane] : 0xFFFFFFFFu;
    const unsigned long long m = __ballot(kj != 0xFFFFFFFFu && (kj >> EMB_ROW_BITS) == id);
    const int cnt = m == ~0ull ? 64 : __ffsll((long long)~m) - 1;        // leading lanes that still belong to the run (sorted: contiguous)
#pragma unroll 8
    for (int t = 0; t < cnt; ++t) {
      const uint32_t r = (uint32_t)__shfl((int)kj, t, 64) & (uint32_t)(EMB_SORT_MAX - 1);
      const float4 v = *(const float4*)(src + (long)r * H);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (cnt < 64) break;
  }
  *(float4*)(table + (long)id * H + chunk * 256 + lane * 4) = acc;
}

// Embedding backward: dx0 (f32 grad of the embedding output) -> LN backward -> scatter-add into the
// word / position tables (float atomics, 256 contiguous bytes per wave instruction), per-block partials
// for LN gamma/beta and the (<= 2 row) token-type table: partials[blk][2 + type_vocab][H].
// row_out != null (dense BERT rows, position id = row % S): the post-LN gradient rows are written there instead of
// being added atomically to dpos (64 samples hitting the same 128 position rows serialise in L2);
// pos_reduce_kernel then sums them over the batch in fixed order.
__global__ __launch_bounds__(256) void embed_bwd_kernel(EmbedArgs a, const float* __restrict__ dx0,
                                                        const float* __restrict__ stats, float* __restrict__ dword,
                                                        float* __restrict__ dpos, float* __restrict__ partials,
                                                        float* __restrict__ row_out, int word_atomic, int pos_atomic) {
  // row_out != null: the post-LayerNorm gradient rows are written there (for pos_reduce_kernel / embed_segsum_kernel);
  // word_atomic / pos_atomic: that table still gets its rows by float atomics (no sorted keys: rows > 8192, or the stand-alone entry point)
  __shared__ float lds[4 * H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Row G = load_row(a.gamma, lane);
  Row accG, accB, accT0, accT1;
#pragma unroll
  for (int i = 0; i < NV; ++i) accG.v[i] = accB.v[i] = accT0.v[i] = accT1.v[i] = float4{0.f, 0.f, 0.f, 0.f};
  for (int r = wave; r < LNB_ROWS; r += 4) {
    const long orow = (long)blockIdx.x * LNB_ROWS + r;
    if (orow >= a.rows) break;
    const long row = a.tok_row ? (long)a.tok_row[orow] : orow;
    if (row < 0) continue;                                  // filler row: no gradient
    int pid, tid_;
    Row XH = embed_gather(a, row, lane, pid, tid_);
    Row DY = load_row(dx0 + orow * H, lane);
    if (a.drop.thresh) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const uint32_t e0 = (uint32_t)(row * H + (i * 64 + lane) * 4);
        float dm[4];
        dropout_mult_n<4>(a.drop, e0, dm);
        DY.v[i].x *= dm[0]; DY.v[i].y *= dm[1]; DY.v[i].z *= dm[2]; DY.v[i].w *= dm[3];
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) { accB.v[i].x += DY.v[i].x; accB.v[i].y += DY.v[i].y; accB.v[i].z += DY.v[i].z; accB.v[i].w += DY.v[i].w; }
    ln_bwd_row(DY, XH, G, stats[orow * 2], stats[orow * 2 + 1]);
#pragma unroll
    for (int i = 0; i < NV; ++i) { accG.v[i].x += XH.v[i].x; accG.v[i].y += XH.v[i].y; accG.v[i].z += XH.v[i].z; accG.v[i].w += XH.v[i].w; }
    long id = a.ids[row];
    id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
    float* wrow = dword + id * H;
    float* prow = dpos + (long)pid * H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (word_atomic) {
        atomicAdd(wrow + c, DY.v[i].x); atomicAdd(wrow + c + 1, DY.v[i].y);
        atomicAdd(wrow + c + 2, DY.v[i].z); atomicAdd(wrow + c + 3, DY.v[i].w);
      }
      if (row_out) *(float4*)(row_out + orow * H + c) = DY.v[i];
      if (pos_atomic) {
        atomicAdd(prow + c, DY.v[i].x); atomicAdd(prow + c + 1, DY.v[i].y);
        atomicAdd(prow + c + 2, DY.v[i].z); atomicAdd(prow + c + 3, DY.v[i].w);
      }
    }
    Row& T = tid_ == 0 ? accT0 : accT1;
#pragma unroll
    for (int i = 0; i < NV; ++i) { T.v[i].x += DY.v[i].x; T.v[i].y += DY.v[i].y; T.v[i].z += DY.v[i].z; T.v[i].w += DY.v[i].w; }
  }
  const int slots = 2 + a.type_vocab;
  float* dst = partials + (long)blockIdx.x * slots * H;
  write_partials(lds, accG, dst, lane, wave);
  write_partials(lds, accB, dst + H, lane, wave);
  write_partials(lds, accT0, dst + 2 * H, lane, wave);
  if (a.type_vocab > 1) write_partials(lds, accT1, dst + 3 * H, lane, wave);
}

__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ partials, float* __restrict__ out,
                                                             int n, int nparts, int accumulate) {
  __shared__ float lds[256];
  int c;
  const float t = partial_colsum16(partials, n, nparts, lds, c, (int)blockIdx.x);
  if (threadIdx.x < PR_COLS && c < n) out[c] = accumulate ? out[c] + t : t;
}

// same reduction, but column c goes to outs.p[c / seg][c % seg] (null pointers are skipped)
__global__ __launch_bounds__(256) void partial_reduce_seg_kernel(const float* __restrict__ partials, SegOuts outs, int seg,
                                                                 int n, int nparts) {
  __shared__ float lds[256];
  int c;
  const float t = partial_colsum16(partials, n, nparts, lds, c, (int)blockIdx.x);
  if (threadIdx.x < PR_COLS && c < n) {
    float* o = outs.p[c / seg];
    if (o) o[c % seg] = t;
  }
}

// Column sums of a bf16 matrix [rows, n] (bias gradients): partials[rowchunk][n], 256 rows per chunk
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, long ld, long rows, int n,
                                                          float* __restrict__ partials) {
  __shared__ float lds[8][256 + 8];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + cg * 8;
  const long r0 = (long)blockIdx.y * 256;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < n) {
    for (long r = r0 + rl; r < r0 + 256 && r < rows; r += 8) {
      const uint4 v = *(const uint4*)(x + r * ld + col);
      acc[0] += bf2f((bf16_t)(v.x & 0xffff)); acc[1] += bf2f((bf16_t)(v.x >> 16));
      acc[2] += bf2f((bf16_t)(v.y & 0xffff)); acc[3] += bf2f((bf16_t)(v.y >> 16));
      acc[4] += bf2f((bf16_t)(v.z & 0xffff)); acc[5] += bf2f((bf16_t)(v.z >> 16));
      acc[6] += bf2f((bf16_t)(v.w & 0xffff)); acc[7] += bf2f((bf16_t)(v.w >> 16));
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) lds[rl][cg * 8 + e] = acc[e];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < n) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += lds[r][c];
    partials[(long)blockIdx.y * n + blockIdx.x * 256 + c] = s;
  }
}

// out[i] = in[idx[i]] (768-wide rows; idx < 0 -> zeros); one wave per row.  f32 and bf16 variants.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ in_f, const bf16_t* __restrict__ in_b,
                                                          const int* __restrict__ idx, int n, float* __restrict__ out_f,
                                                          bf16_t* __restrict__ out_b) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const long r = idx[i];
  if (in_f) {
    Row X;
    if (r >= 0) X = load_row(in_f + r * H, lane);
    else for (int v = 0; v < NV; ++v) X.v[v] = float4{0.f, 0.f, 0.f, 0.f};
    store_row(out_f + (long)i * H, lane, X);
  }
  if (in_b) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * 64 + lane) * 4;
      uint2 x = {0u, 0u};
      if (r >= 0) x = *(const uint2*)(in_b + r * H + c);
      *(uint2*)(out_b + (long)i * H + c) = x;
    }
  }
}
// out[idx[i]] = in[i] for idx[i] >= 0 (destination pre-zeroed by the caller)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ in_f, const bf16_t* __restrict__ in_b,
                                                           const int* __restrict__ idx, int n, float* __restrict__ out_f,
                                                           bf16_t* __restrict__ out_b) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const long r = idx[i];
  if (r < 0) return;
  if (in_f) store_row(out_f + r * H, lane, load_row(in_f + (long)i * H, lane));
  if (in_b) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * 64 + lane) * 4;
      *(uint2*)(out_b + r * H + c) = *(const uint2*)(in_b + (long)i * H + c);
    }
  }
}

int gather_rows(const void* in_f32, const void* in_bf16, const void* idx, int n, void* out_f32, void* out_bf16, hipStream_t stream) {
  hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, (const float*)in_f32, (const bf16_t*)in_bf16,
                     (const int*)idx, n, (float*)out_f32, (bf16_t*)out_bf16);
  return check_launch("gather_rows_kernel");
}
int scatter_rows(const void* in_f32, const void* in_bf16, const void* idx, int n, void* out_f32, void* out_bf16, hipStream_t stream) {
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, (const float*)in_f32, (const bf16_t*)in_bf16,
                     (const int*)idx, n, (float*)out_f32, (bf16_t*)out_bf16);
  return check_launch("scatter_rows_kernel");
}

}  // namespace carel

using namespace carel;

static EmbedArgs make_embed(const carel_embed_args* a) {
  EmbedArgs e;
  e.ids = (const long*)a->input_ids; e.tt = (const long*)a->token_type_ids;
  e.word = (const float*)a->word_emb; e.pos = (const float*)a->pos_emb; e.type = (const float*)a->type_emb;
  e.gamma = (const float*)a->ln_gamma; e.beta = (const float*)a->ln_beta;
  e.eps = a->ln_eps; e.S = a->seq_len; e.roberta = a->roberta; e.pad_id = a->pad_id;
  e.tok_row = (const int*)a->tok_row;
  e.rows = (a->tok_row && a->n_rows > 0) ? (long)a->n_rows : (long)a->batch * a->seq_len;
  e.vocab = a->vocab_size; e.max_pos = a->max_pos; e.type_vocab = a->type_vocab;
  e.drop = make_dropout(a->drop_seed, 0u, a->drop_p, a->drop_idx_offset);
  return e;
}

static int embed_check(const carel_embed_args* a, const char* who) {
  if (!a) return set_error(CAREL_ERR_ARG, "%s: null args", who);
  if (a->hidden != H) return set_error(CAREL_ERR_SHAPE, "%s: hidden must be %d (got %d)", who, H, a->hidden);
  if (!a->input_ids || !a->word_emb || !a->pos_emb || !a->type_emb || !a->ln_gamma || !a->ln_beta)
    return set_error(CAREL_ERR_ARG, "%s: null tensor", who);
  if (a->batch <= 0 || a->seq_len <= 0 || a->type_vocab < 1 || a->type_vocab > 2 || a->seq_len > a->max_pos)
    return set_error(CAREL_ERR_SHAPE, "%s: bad batch/seq_len/type_vocab", who);
  return CAREL_OK;
}

extern "C" int carel_embed_ln_fwd(const carel_embed_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = embed_check(a, "carel_embed_ln_fwd");
  if (rc) return rc;
  if (!a->x_f32 || !a->x_bf16 || !a->stats) return set_error(CAREL_ERR_ARG, "carel_embed_ln_fwd: null output");
  EmbedArgs e = make_embed(a);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)((e.rows + 3) / 4)), dim3(256), 0, stream, e, (float*)a->x_f32,
                     (bf16_t*)a->x_bf16, (float*)a->stats, (int2*)nullptr);
  return check_launch("embed_fwd_kernel");
}

namespace carel {
// can the table gradients of this batch come from sorted keys (embed_sort_kernel's limits)?
int embed_sort_supported(const carel_embed_args* a) {
  const long rows = (a->tok_row && a->n_rows > 0) ? (long)a->n_rows : (long)a->batch * a->seq_len;
  return rows >= 1 && rows <= EMB_SORT_MAX && a->vocab_size <= (1 << (32 - EMB_ROW_BITS)) && a->max_pos <= (1 << (32 - EMB_ROW_BITS));
}
size_t embed_sort_bytes() { return (size_t)EMB_SORT_MAX * 8 + 2 * (size_t)EMB_SORT_MAX * 4; }      // row keys int2 [8192] | sorted keys u32 [2][8192]
// carel_embed_ln_fwd that also leaves the (token id, position id) of every row in `sort_ws`; embed_sort_rows then sorts them (any stream
// ordered after this call)
int embed_ln_fwd_keys(const carel_embed_args* a, void* sort_ws, hipStream_t stream) {
  int rc = embed_check(a, "carel_embed_ln_fwd");
  if (rc) return rc;
  if (!a->x_f32 || !a->x_bf16 || !a->stats || !sort_ws) return set_error(CAREL_ERR_ARG, "carel_embed_ln_fwd: null output");
  EmbedArgs e = make_embed(a);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)((e.rows + 3) / 4)), dim3(256), 0, stream, e, (float*)a->x_f32,
                     (bf16_t*)a->x_bf16, (float*)a->stats, (int2*)sort_ws);
  return check_launch("embed_fwd_kernel");
}
int embed_sort_rows(const carel_embed_args* a, void* sort_ws, hipStream_t stream) {
  const long rows = (a->tok_row && a->n_rows > 0) ? (long)a->n_rows : (long)a->batch * a->seq_len;
  hipLaunchKernelGGL(embed_sort_kernel, dim3(2), dim3(1024), 0, stream, (const int2*)sort_ws, (int)rows, (uint32_t*)((char*)sort_ws + (size_t)EMB_SORT_MAX * 8));
  return check_launch("embed_sort_kernel");
}
}  // namespace carel

extern "C" int carel_embed_ln_bwd_blocks(int64_t rows) { return (int)((rows + LNB_ROWS - 1) / LNB_ROWS); }

namespace carel {
// row_scratch: optional f32 [rows, hidden] buffer; with it, dense BERT batches get their position-table gradient from
// a fixed-order reduction instead of float atomics (see embed_bwd_kernel)
// sort_ws (with row_scratch): the workspace embed_ln_fwd_keys / embed_sort_rows filled for THIS batch -- both table gradients then come
// from fixed-order sums (embed_segsum_kernel; dense BERT positions keep pos_reduce_kernel): no atomics anywhere, bit-reproducible
int embed_ln_bwd_ex(const carel_embed_args* a, const void* dx0, void* dword, void* dpos, void* dtype_, void* dgamma, void* dbeta,
                    void* partials, void* row_scratch, hipStream_t stream, const void* sort_ws) {
  int rc = embed_check(a, "carel_embed_ln_bwd");
  if (rc) return rc;
  if (!dx0 || !dword || !dpos || !dtype_ || !dgamma || !dbeta || !partials || !a->stats)
    return set_error(CAREL_ERR_ARG, "carel_embed_ln_bwd: null tensor");
  EmbedArgs e = make_embed(a);
  const int nblk = carel_embed_ln_bwd_blocks(e.rows);
  const int slots = 2 + e.type_vocab;
  const bool by_reduction = row_scratch && !e.roberta && !e.tok_row && e.rows == (long)a->batch * a->seq_len;
  const bool sorted = row_scratch && sort_ws && embed_sort_supported(a);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(nblk), dim3(256), 0, stream, e, (const float*)dx0, (const float*)a->stats,
                     (float*)dword, (float*)dpos, (float*)partials, (by_reduction || sorted) ? (float*)row_scratch : (float*)nullptr,
                     sorted ? 0 : 1, (by_reduction || sorted) ? 0 : 1);
  rc = check_launch("embed_bwd_kernel");
  if (rc) return rc;
  if (by_reduction) {
    hipLaunchKernelGGL(pos_reduce_kernel, dim3((unsigned)(((long)a->seq_len * H / 4 + 255) / 256)), dim3(256), 0, stream,
                       (const float*)row_scratch, a->batch, a->seq_len, (float*)dpos);
    if ((rc = check_launch("pos_reduce_kernel"))) return rc;
  }
  if (sorted) {
    const uint32_t* keys = (const uint32_t*)((const char*)sort_ws + (size_t)EMB_SORT_MAX * 8);
    const unsigned blocks = (unsigned)((EMB_SORT_MAX * 3 + 3) / 4);
    hipLaunchKernelGGL(embed_segsum_kernel, dim3(blocks), dim3(256), 0, stream, keys, (const float*)row_scratch, (float*)dword, a->vocab_size);
    if ((rc = check_launch("embed_segsum_kernel"))) return rc;
    if (!by_reduction) {
      hipLaunchKernelGGL(embed_segsum_kernel, dim3(blocks), dim3(256), 0, stream, keys + EMB_SORT_MAX, (const float*)row_scratch, (float*)dpos, a->max_pos);
      if ((rc = check_launch("embed_segsum_kernel"))) return rc;
    }
  }
  SegOuts so; so.p[0] = (float*)dgamma; so.p[1] = (float*)dbeta; so.p[2] = (float*)dtype_;
  so.p[3] = e.type_vocab > 1 ? (float*)dtype_ + H : nullptr;
  hipLaunchKernelGGL(partial_reduce_seg_kernel, dim3((slots * H + PR_COLS - 1) / PR_COLS), dim3(256), 0, stream, (const float*)partials,
                     so, H, slots * H, nblk);
  return check_launch("partial_reduce_seg_kernel");
}
}  // namespace carel

extern "C" int carel_embed_ln_bwd(const carel_embed_args* a, const void* dx0, void* dword, void* dpos, void* dtype_,
                                  void* dgamma, void* dbeta, void* partials, void* stream_) {
  return embed_ln_bwd_ex(a, dx0, dword, dpos, dtype_, dgamma, dbeta, partials, nullptr, (hipStream_t)stream_, nullptr);
}

extern "C" int carel_layernorm_fwd(const void* h, const void* gamma, const void* beta, float eps, int64_t rows, int32_t hidden,
                                   void* x_f32, void* x_bf16, void* stats, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (hidden != H) return set_error(CAREL_ERR_SHAPE, "carel_layernorm_fwd: hidden must be %d", H);
  if (!h || !gamma || !beta || rows <= 0) return set_error(CAREL_ERR_ARG, "carel_layernorm_fwd: bad arguments");
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const float*)h, (const float*)gamma,
                     (const float*)beta, eps, (long)rows, (float*)x_f32, (bf16_t*)x_bf16, (float*)stats);
  return check_launch("ln_fwd_kernel");
}

// Rows per wave of the LayerNorm backward row kernel: 4 (16 rows per workgroup) fills the chip at the dense 8192 rows (512 workgroups); packed ECPE batches
// (~1.7 k rows) gave 104 workgroups, each wave walking four rows' loads -> reductions -> stores in turn: one row per wave there (416 workgroups,
// `ln_bwd_kernel<true>` 15 -> 9-10 us), two up to 4096 rows.  More blocks = more partial rows for the reduction behind it (at most 512 up to 8192 rows).
static int ln_bwd_rpw(int64_t rows) { return rows <= 2048 ? 1 : (rows <= 4096 ? 2 : 4); }
extern "C" int carel_layernorm_bwd_blocks(int64_t rows) { const int wg = 4 * ln_bwd_rpw(rows); return (int)((rows + wg - 1) / wg); }
namespace carel {
// the most blocks ANY row count <= rows can need (scratch sizing: a packed batch has fewer rows than batch x seq_len, and may need more blocks)
int layernorm_bwd_blocks_max(int64_t rows) {
  if (rows <= 2048) return (int)((rows + 3) / 4);
  const int64_t big = (rows + 15) / 16;
  return (int)(big > 512 ? big : 512);
}
}

// partials: f32 scratch of carel_layernorm_bwd_blocks(rows) * 3 * hidden floats
extern "C" int carel_layernorm_bwd(const void* dy, const void* h, const void* stats, const void* gamma, int64_t rows,
                                   int32_t hidden, uint32_t drop_seed, uint32_t drop_site, uint32_t drop_idx_offset,
                                   float drop_p, void* dh_f32, void* dy_bf16, void* dgamma, void* dbeta, void* dbias,
                                   void* partials, void* stream_) {
  return carel_layernorm_bwd_packed(dy, h, stats, gamma, rows, hidden, drop_seed, drop_site, drop_idx_offset, drop_p, nullptr, dh_f32,
                                    dy_bf16, dgamma, dbeta, dbias, partials, stream_);
}

// internal: the row kernel on `stream`, the reduction of its partials on `reduce_stream` (the caller orders the two
// streams: encoder.hip forks the side stream between the calls); reduce_stream == stream is the plain serial form.
namespace carel {
int layernorm_bwd_rows(const void* dy, const void* h, const void* stats, const void* gamma, int64_t rows, uint32_t drop_seed, uint32_t drop_site,
                       uint32_t drop_idx_offset, float drop_p, const void* drop_row_map, void* dh_f32, void* dy_bf16, void* partials,
                       hipStream_t stream) {
  if (!dy || !h || !stats || !gamma || !partials || rows <= 0) return set_error(CAREL_ERR_ARG, "carel_layernorm_bwd: bad arguments");
  const int nblk = carel_layernorm_bwd_blocks(rows);
  const Dropout drop = make_dropout(drop_seed, drop_site, drop_p, drop_idx_offset);
  const LnSlabSrc src{1, 0, nullptr};
#define CAREL_LNB(R) hipLaunchKernelGGL((ln_bwd_kernel<false, R>), dim3(nblk), dim3(256), 0, stream, (const float*)dy, (const float*)h, (const float*)stats, \
                     (const float*)gamma, (long)rows, drop, (const int*)drop_row_map, (float*)dh_f32, (bf16_t*)dy_bf16, (float*)partials, src)
  switch (ln_bwd_rpw(rows)) { case 1: CAREL_LNB(1); break; case 2: CAREL_LNB(2); break; default: CAREL_LNB(4); }
#undef CAREL_LNB
  return check_launch("ln_bwd_kernel");
}
int layernorm_bwd_rows_slabs(const void* slabs, int splits, const void* resid, const void* h, const void* stats, const void* gamma, int64_t rows,
                             uint32_t drop_seed, uint32_t drop_site, uint32_t drop_idx_offset, float drop_p, const void* drop_row_map, void* dh_f32,
                             void* dy_bf16, void* partials, hipStream_t stream) {
  if (!slabs || splits < 2 || !h || !stats || !gamma || !partials || rows <= 0) return set_error(CAREL_ERR_ARG, "layernorm_bwd_rows_slabs: bad arguments");
  const int nblk = carel_layernorm_bwd_blocks(rows);
  const Dropout drop = make_dropout(drop_seed, drop_site, drop_p, drop_idx_offset);
  const LnSlabSrc src{splits, (long)rows * H, (const float*)resid};
#define CAREL_LNB(R) hipLaunchKernelGGL((ln_bwd_kernel<true, R>), dim3(nblk), dim3(256), 0, stream, (const float*)slabs, (const float*)h, (const float*)stats, \
                     (const float*)gamma, (long)rows, drop, (const int*)drop_row_map, (float*)dh_f32, (bf16_t*)dy_bf16, (float*)partials, src)
  switch (ln_bwd_rpw(rows)) { case 1: CAREL_LNB(1); break; case 2: CAREL_LNB(2); break; default: CAREL_LNB(4); }
#undef CAREL_LNB
  return check_launch("ln_bwd_kernel<slabs>");
}
int layernorm_fwd_slabs(const void* slabs, int splits, const void* bias, const void* resid, const void* resid_stats, const void* resid_gamma,
                        const void* resid_beta, uint32_t drop_seed, uint32_t drop_site, uint32_t drop_idx_offset, float drop_p, const void* drop_row_map,
                        void* h_out, const void* gamma, const void* beta, float eps, int64_t rows, void* x_f32, void* x_bf16, void* stats,
                        hipStream_t stream) {
  if (!slabs || splits < 2 || !resid || !gamma || !beta || rows <= 0) return set_error(CAREL_ERR_ARG, "layernorm_fwd_slabs: bad arguments");
  if (resid_stats && (!resid_gamma || !resid_beta)) return set_error(CAREL_ERR_ARG, "layernorm_fwd_slabs: recomputed residual needs gamma and beta");
  LnFwdSlabs a;
  a.slabs = (const float*)slabs; a.splits = splits; a.plane = (long)rows * H; a.bias = (const float*)bias; a.resid = (const float*)resid;
  a.resid_stats = (const float*)resid_stats; a.resid_gamma = (const float*)resid_gamma; a.resid_beta = (const float*)resid_beta;
  a.drop = make_dropout(drop_seed, drop_site, drop_p, drop_idx_offset); a.row_map = (const int*)drop_row_map; a.h_out = (float*)h_out;
  hipLaunchKernelGGL(ln_fwd_slabs_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a, (const float*)gamma, (const float*)beta, eps, (long)rows,
                     (float*)x_f32, (bf16_t*)x_bf16, (float*)stats);
  return check_launch("ln_fwd_slabs_kernel");
}
int layernorm_bwd_reduce(const void* partials, int64_t rows, void* dgamma, void* dbeta, void* dbias, hipStream_t stream) {
  const int nblk = carel_layernorm_bwd_blocks(rows);
  SegOuts so; so.p[0] = (float*)dgamma; so.p[1] = (float*)dbeta; so.p[2] = (float*)dbias; so.p[3] = nullptr;
  hipLaunchKernelGGL(partial_reduce_seg_kernel, dim3((3 * H + PR_COLS - 1) / PR_COLS), dim3(256), 0, stream, (const float*)partials, so,
                     H, 3 * H, nblk);
  return check_launch("partial_reduce_seg_kernel");
}
}  // namespace carel

extern "C" int carel_layernorm_bwd_packed(const void* dy, const void* h, const void* stats, const void* gamma, int64_t rows,
                                          int32_t hidden, uint32_t drop_seed, uint32_t drop_site, uint32_t drop_idx_offset,
                                          float drop_p, const void* drop_row_map, void* dh_f32, void* dy_bf16, void* dgamma,
                                          void* dbeta, void* dbias, void* partials, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (hidden != H) return set_error(CAREL_ERR_SHAPE, "carel_layernorm_bwd: hidden must be %d", H);
  int rc = layernorm_bwd_rows(dy, h, stats, gamma, rows, drop_seed, drop_site, drop_idx_offset, drop_p, drop_row_map, dh_f32, dy_bf16,
                              partials, stream);
  if (rc) return rc;
  return layernorm_bwd_reduce(partials, rows, dgamma, dbeta, dbias, stream);
}

extern "C" int carel_colsum_bf16(const void* x, int64_t ld, int64_t rows, int32_t n, void* out_f32, int32_t accumulate,
                                 void* partials, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out_f32 || !partials || rows <= 0 || n <= 0 || (n & 7) || (ld & 7))
    return set_error(CAREL_ERR_ARG, "carel_colsum_bf16: bad arguments (n, ld multiples of 8)");
  const int chunks = (int)((rows + 255) / 256);
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((n + 255) / 256, chunks), dim3(256), 0, stream, (const bf16_t*)x, (long)ld,
                     (long)rows, n, (float*)partials);
  int rc = check_launch("colsum_bf16_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((n + PR_COLS - 1) / PR_COLS), dim3(256), 0, stream, (const float*)partials, (float*)out_f32, n,
                     chunks, accumulate);
  return check_launch("partial_reduce_kernel");
}

extern "C" int carel_partial_reduce_f32(const void* partials, void* out, int32_t n, int32_t nparts, int32_t accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!partials || !out || n <= 0 || nparts <= 0) return set_error(CAREL_ERR_ARG, "carel_partial_reduce_f32: bad arguments");
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((n + PR_COLS - 1) / PR_COLS), dim3(256), 0, stream, (const float*)partials, (float*)out, n, nparts, accumulate);
  return check_launch("partial_reduce_kernel");
}
