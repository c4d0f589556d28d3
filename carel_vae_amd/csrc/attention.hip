// Self-attention forward / backward for one (sample, head) per workgroup, S <= 128, head_dim = 64.
// Replaces transformers BertSelfAttention (eager_attention_forward: QK^T * scale + additive key mask ->
// softmax -> dropout -> P V) and its autograd backward; reached from
// drl_classifier_ec_mmd_final_mul.py:202-206 / :841.
//
// qkv is the fused projection output [T, 3*768] bf16 (q | k | v), head h at columns h*64.
// Forward  : per wave 32 queries.  S^T = K Q^T on v_mfma_f32_32x32x16_bf16 with the KEY on the accumulator
//            row (registers) and the QUERY on the lane, so softmax over keys is in-lane + one half swap,
//            and the probability tile is directly the B operand of O^T = V^T P^T (no LDS round trip).
// Backward : per wave 32 keys (key on the lane).  S and dP accumulators are directly the B operands of
//            dV^T = dO^T P and dK^T = Q^T dS; dS^T crosses LDS once for dQ^T = K^T dS^T.
// K/V/Q/dO tiles are LDS images with 128-B rows filled by global_load_lds_dwordx4; the XOR swizzle
// f_att serves both the row reads (ds_read_b128) and the transposed reads (ds_read_b64_tr_b16).
#include "carel_hip_internal.h"

namespace carel {

constexpr int HD = 64;        // head dim
constexpr int NH = 12;        // heads
constexpr int HID = NH * HD;  // 768
constexpr int QKV_LD = 3 * HID;

__device__ __forceinline__ int f_att(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int att_off(int row, int chunk) { return row * 128 + ((chunk ^ f_att(row)) << 4); }

// fill an image of `rows` x 64 bf16 from a row-major global matrix (row stride ld elements); 256 threads
__device__ __forceinline__ void stage_att(const bf16_t* __restrict__ g, long ld, int rows, char* img) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int q = wave; q < (rows >> 3); q += 4) {
    const int r = q * 8 + (lane >> 3);
    const int c = (lane & 7) ^ f_att(r);
    __builtin_amdgcn_global_load_lds(g + (long)r * ld + c * 8, (CAREL_LDS void*)(img + q * 1024), 16, 0, 0);
  }
}

// 32x32x16 operand whose own-matrix row is on the lane: X[row = r0 + (l&31)][kk = 16*s + 8*(l>>5) + j]
__device__ __forceinline__ bf16x8 frag32_row(const char* img, int r0, int s) {
  const int l = threadIdx.x & 63;
  return *(const bf16x8*)(img + att_off(r0 + (l & 31), 2 * s + (l >> 5)));
}
// 32x32x16 operand read TRANSPOSED from an image M[kk][x]: lane holds M[kk(j)][x = x0 + (l&31)].
//   PERM = false: kk(j) = kb + 8*(l>>5) + j                      (natural order)
//   PERM = true : kk(j) = kb + 8*(j>>2) + 4*(l>>5) + (j&3)        (pairs with an accumulator tile used as
//                                                                  the other operand, carel_common.h)
template <bool PERM>
__device__ __forceinline__ bf16x8 frag32_tr(const char* img, int x0, int kb) {
  const int l = threadIdx.x & 63;
  const int g = l >> 4, hh = g >> 1, qq = (l & 15) >> 2, p = l & 3;
  const int chunk = ((x0 + 16 * (g & 1)) >> 3) + (p >> 1), sub = (p & 1) * 8;
  const int r0 = PERM ? (kb + 4 * hh + qq) : (kb + 8 * hh + qq);
  const int r1 = PERM ? (r0 + 8) : (r0 + 4);
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((CAREL_LDS s16x4*)(img + att_off(r0, chunk) + sub));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((CAREL_LDS s16x4*)(img + att_off(r1, chunk) + sub));
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// registers 8s..8s+7 of a 32x32 accumulator as the bf16 B operand of k-step s (rows of the tile = kk)
__device__ __forceinline__ bf16x8 acc_as_operand(const f32x16& x, int s) {
  const uint4 r = {pack2bf(x[8 * s], x[8 * s + 1]), pack2bf(x[8 * s + 2], x[8 * s + 3]), pack2bf(x[8 * s + 4], x[8 * s + 5]), pack2bf(x[8 * s + 6], x[8 * s + 7])};
  return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ bf16x8 load_frag_global(const bf16_t* p) { return *(const bf16x8*)p; }


// A wave's [32 rows][64 d] result sits in two 32x32 accumulators with the ROW on the lane and 4-element groups of d spread over
// the registers and the two half-waves: stored straight from there every instruction writes 16-byte fragments of 32 different
// rows (8 instructions per 128-byte line; measured: the stores were 27 % of the backward kernel).  Through a 4-KiB LDS slot of the
// wave's own (XOR-swizzled 16-byte chunks: conflict-free both ways) every instruction stores 8 whole 128-byte rows instead.
__device__ __forceinline__ void store_rows_via_lds(const f32x16 (&acc)[2], float scale, char* slot, bf16_t* grow0, long ld, int row_base, int nrows_live) {
  const int l = threadIdx.x & 63, r = l & 31, hh = l >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = dt * 32 + 8 * i + 4 * hh;                          // 4 consecutive d
      const int chunk = (d >> 3) ^ ((r >> 1) & 7);
      const uint2 v = {pack2bf(acc[dt][4 * i] * scale, acc[dt][4 * i + 1] * scale), pack2bf(acc[dt][4 * i + 2] * scale, acc[dt][4 * i + 3] * scale)};
      *(uint2*)(slot + r * 128 + chunk * 16 + (d & 4) * 2) = v;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // the slot is this wave's own: no barrier needed
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (l >> 3), c = l & 7;
    const uint4 v = *(const uint4*)(slot + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
    if (row_base + row < nrows_live) *(uint4*)(grow0 + (long)row * ld + c * 8) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // reads done before the slot is rewritten
}

struct AttnParams {
  const bf16_t* qkv;        // [B*S, 2304]
  const long* att_mask;     // [B, S] (1 = attend) or null
  bf16_t* ctx;              // fwd out / bwd in  [B*S, 768]
  float* lse;               // [B, NH, S]
  const bf16_t* dctx;       // bwd in   [B*S, 768]
  bf16_t* dqkv;             // bwd out  [B*S, 2304]
  int B, S;
  Dropout drop;             // element index ((b*NH + h)*S + q)*S + k
  const int* cu;            // packed: rows [cu[b], cu[b+1]) belong to sample b (null = dense, rows b*S ..)
  const float* rel;         // MPNet relative-position bias by distance: [NH][256], entry 127 + (key - query); null = none
  int qlim;                 // 0 = all; else only the first qlim (multiple of 32) positions of every sample are live queries
  float* drel;              // bwd: its gradient by distance, [B * NH][256]: row (sample, head) is read-modify-written by that workgroup alone (every
                            // layer adds to it in stream order) -- no atomics anywhere, so the table gradient is bit-reproducible
};

#ifndef CAREL_ATTN_ABLATE
#define CAREL_ATTN_ABLATE 0        // timing ablations of the backward kernel (wrong results): tools/ablate_attn.py
#endif
constexpr float MASK_NEG = -3.4028234663852886e38f;   // torch.finfo(float32).min, as HF adds it
constexpr int ATTN_BWD_LDS = 16384 + 16384 + 32768 + 1024;
constexpr int ATTN_BWD_LDS_REL = ATTN_BWD_LDS + 1024 + 4096;     // + bias by distance [256] + its gradient, one array per wave [4][256]

// =========================================================================================== forward
// REL: scores += rel[h][127 + key - query] (transformers MPNetAttention: `attention_scores += position_bias`, the bias shared by all
// layers; en_ec_sentence_transformer.py:22 loads all-mpnet-base-v2)
template <bool REL, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 16384 + 512 + (REL ? 1024 : 0)];
  char* kimg = smem;
  char* vimg = smem + 16384;
  float* maskadd = (float*)(smem + 32768);
  float* relb = (float*)(smem + 32768 + 512);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / NH, h = blockIdx.x - b * NH;
  const int S = p.S;
  // packed: this sample's `len` tokens start at row cu[b]; tiles may run past them into rows of the next sample
  // (finite data, masked as keys, never stored as queries)
  const long row0 = p.cu ? (long)p.cu[b] : (long)b * S;
  const int len = p.cu ? (p.cu[b + 1] - p.cu[b]) : S;
  const int nkt = (len + 31) >> 5, rows = nkt << 5;
  const bf16_t* qbase = p.qkv + row0 * QKV_LD + h * HD;
  stage_att(qbase + HID, QKV_LD, rows, kimg);
  stage_att(qbase + 2 * HID, QKV_LD, rows, vimg);
  int masked_here = 0;
  for (int k = threadIdx.x; k < rows; k += 256) {
    const float ma = p.cu ? (k < len ? 0.f : MASK_NEG) : ((p.att_mask && p.att_mask[row0 + k] == 0) ? MASK_NEG : 0.f);
    maskadd[k] = ma;
    masked_here |= ma != 0.f;
  }
  if (REL) relb[threadIdx.x] = p.rel[h * 256 + threadIdx.x];
  const int q0 = wave * 32;
  const int hh = lane >> 5;
  // Q fragments (B operand of S^T = K Q^T): Q[q0 + (l&31)][16s + 8hh + j] -- requested before the wait for the staged K / V images
  bf16x8 qf[4];
  const bool qact = wave < nkt && (p.qlim == 0 || q0 < p.qlim);       // wave-uniform: this wave's 32 queries exist and are wanted
  if (qact) {
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = load_frag_global(qbase + (long)(q0 + (lane & 31)) * QKV_LD + 16 * s + 8 * hh);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // (the barrier also tells every wave whether ANY key of this sample is masked: dense batches without padding -- the bench shape -- and
  // every key tile of a packed sample but a ragged last one then skip the mask term's LDS read and add: 64 of each per lane)
  const int any_masked = __syncthreads_or(masked_here);
  if (!qact) return;                             // no barrier below this point

  f32x16 x[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x[kt][r] = 0.f;
    if (kt < nkt) {
#pragma unroll
      for (int s = 0; s < 4; ++s) x[kt] = mfma32(frag32_row(kimg, kt * 32, s), qf[s], x[kt]);
    }
  }
  // softmax over keys: this lane holds, for query q0+(l&31), keys kt*32 + acc32_row(r, lane).  The scores are kept in the log2 domain
  // (scale and log2(e) in one multiply, exp2 instead of exp: one VALU instruction less per element; the backward pass has always recomputed the
  // probabilities that way); a masked key adds finfo.min exactly as HF does, so a sample with no attended key still gets the uniform row.
  constexpr float SC2 = 0.125f * 1.4426950408889634f;
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt < nkt) {
      // wave-uniform: does this key tile hold a masked key?  (packed: only a ragged last tile; dense: unknown per tile, any tile may)
      const bool tile_masked = p.cu ? (kt == nkt - 1 && (len & 31) != 0) : (any_masked != 0);
      if (tile_masked) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + acc32_row(r, lane);
          float v = fmaf(x[kt][r], SC2, maskadd[key]);
          if (REL) v = fmaf(relb[127 + key - (q0 + (lane & 31))], 1.4426950408889634f, v);
          x[kt][r] = v;
          m = fmaxf(m, v);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = x[kt][r] * SC2;
          if (REL) v = fmaf(relb[127 + kt * 32 + acc32_row(r, lane) - (q0 + (lane & 31))], 1.4426950408889634f, v);
          x[kt][r] = v;
          m = fmaxf(m, v);
        }
      }
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float lsum = 0.f;
  // dropout_hash2's argument for key 0 (32-bit wrap-around arithmetic, as the element index is defined); keys kt*32 + acc32_row(r) are added per pair
  const uint32_t ebase = (uint32_t)((((long)b * NH + h) * S + (q0 + (lane & 31))) * S) + (uint32_t)(4 * hh) + p.drop.idx_offset;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt < nkt) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {          // keys acc32_row(r), acc32_row(r + 1) are an aligned pair: one hash (ebase, S and the offset are even)
        const float e0 = __builtin_amdgcn_exp2f(x[kt][r] - m), e1 = __builtin_amdgcn_exp2f(x[kt][r + 1] - m);
        lsum += e0; lsum += e1;
        float d0 = 1.0f, d1 = 1.0f;
        if constexpr (DROP) {       // (a template parameter: tested at run time, every pair sat in its own basic block)
          const uint32_t hsh = mix32(((ebase + (uint32_t)(kt * 32 + (r & 3) + 8 * (r >> 2))) >> 1) ^ p.drop.key);
          d0 = dropout_pick(p.drop, hsh, 0u); d1 = dropout_pick(p.drop, hsh, 1u);
        }
        x[kt][r] = e0 * d0; x[kt][r + 1] = e1 * d1;
      }
    }
  }
  lsum += __shfl_xor(lsum, 32, 64);
  const bool qlive = q0 + (lane & 31) < len;
  if (hh == 0 && qlive) p.lse[((long)b * NH + h) * S + q0 + (lane & 31)] = (m + __builtin_amdgcn_logf(lsum)) * 0.6931471805599453f;      // natural log-sum-exp, as before
  const float inv = 1.0f / lsum;
  // O^T[d][q] = sum_k V^T[d][k] P^T[k][q]
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt < nkt) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = acc_as_operand(x[kt], s);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(frag32_tr<true>(vimg, dt * 32, kt * 32 + 16 * s), pf, o[dt]);
      }
    }
  }
  // (whole-row stores through an extra 16 KiB of LDS were measured here too: no gain -- 12.6 MB of output against 37.7 MB of input)
  bf16_t* crow = p.ctx + (row0 + q0 + (lane & 31)) * HID + h * HD;
  if (qlive) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = dt * 32 + 8 * i + 4 * hh;
        uint2 v = {pack2bf(o[dt][4 * i] * inv, o[dt][4 * i + 1] * inv), pack2bf(o[dt][4 * i + 2] * inv, o[dt][4 * i + 3] * inv)};
        *(uint2*)(crow + d) = v;
      }
  }
}

// =========================================================================================== backward
__device__ __forceinline__ int swz_ds(int k) { return ((k & 3) << 3) | ((k >> 2) & 7); }
__device__ __forceinline__ int ds_off(int k, int q) { return k * 256 + ((((q >> 2) ^ swz_ds(k)) & 31) << 3) + (q & 3) * 2; }

// DROP (attention-probability dropout on) is a template parameter, and the per-element arithmetic below is straight-line code: with the
// rate tested at run time and `live ? exp(..) : 0` written as conditionals, the compiler (ROCm 7.2) built 48 branches per query tile, each
// around one ds_read_b32 of lse[q] / delta[q] with its own lgkmcnt(0) -- 32 exposed LDS round trips per tile and wave.  Rows past the sample
// get lse = +inf instead (exp2(-inf) = 0: the exact zeros the `live` test produced) and masked / past-the-sample keys -inf through the mask term.
constexpr float LOG2E = 1.4426950408889634f;
template <bool REL, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // ATTN_BWD_LDS (+ REL: 2048) bytes
  char* qimg = smem;                 // Q  [S][64]
  char* doimg = smem + 16384;        // dO [S][64]; re-used for K in the dQ phase
  char* dsimg = smem + 32768;        // dS^T [k][q] bf16, 256-B rows
  float* lse = (float*)(smem + 65536);
  float* delta = lse + 128;
  float* relb = (float*)(smem + ATTN_BWD_LDS);                  // REL: bias by distance, and the sums of dS by distance, one private array per wave
  float* relg = relb + 256;                                     // [4][256]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / NH, h = blockIdx.x - b * NH;
  const int S = p.S;
  const long row0 = p.cu ? (long)p.cu[b] : (long)b * S;
  const int len = p.cu ? (p.cu[b + 1] - p.cu[b]) : S;
  const int nt = (len + 31) >> 5, rows = nt << 5;
  const int ntq = p.qlim ? min(nt, p.qlim >> 5) : nt;         // query tiles that carry a gradient (q_rows: the rest have dctx = 0)
  const bf16_t* qbase = p.qkv + row0 * QKV_LD + h * HD;
  const bf16_t* dobase = p.dctx + row0 * HID + h * HD;
  const bf16_t* obase = p.ctx + row0 * HID + h * HD;
  stage_att(qbase, QKV_LD, rows, qimg);
  stage_att(dobase, HID, rows, doimg);
  if (REL) {
    relb[threadIdx.x] = p.rel[h * 256 + threadIdx.x];
#pragma unroll
    for (int w = 0; w < 4; ++w) relg[w * 256 + threadIdx.x] = 0.f;
  }
  // lse in log2 units; +inf for the rows past the sample (they belong to its neighbours: probability exactly 0).  (The clamp keeps a
  // fully masked row -- lse = -3.4e38 -- finite: its probabilities come out 0 here; no input of the reference has such a row.)
  for (int k = threadIdx.x; k < rows; k += 256) lse[k] = k < len ? fmaxf(p.lse[((long)b * NH + h) * S + k], -1e30f) * LOG2E : INFINITY;
  {  // delta[q] = sum_d dO[q][d] * O[q][d]; 2 threads per query, 32 d each
    const int q = threadIdx.x >> 1, half = threadIdx.x & 1;
    float s = 0.f;
    if (q < len) {
      const uint4* a = (const uint4*)(dobase + (long)q * HID + half * 32);
      const uint4* c = (const uint4*)(obase + (long)q * HID + half * 32);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4 u = a[i], v = c[i];
        s += bf2f((bf16_t)(u.x & 0xffff)) * bf2f((bf16_t)(v.x & 0xffff)) + bf2f((bf16_t)(u.x >> 16)) * bf2f((bf16_t)(v.x >> 16));
        s += bf2f((bf16_t)(u.y & 0xffff)) * bf2f((bf16_t)(v.y & 0xffff)) + bf2f((bf16_t)(u.y >> 16)) * bf2f((bf16_t)(v.y >> 16));
        s += bf2f((bf16_t)(u.z & 0xffff)) * bf2f((bf16_t)(v.z & 0xffff)) + bf2f((bf16_t)(u.z >> 16)) * bf2f((bf16_t)(v.z >> 16));
        s += bf2f((bf16_t)(u.w & 0xffff)) * bf2f((bf16_t)(v.w & 0xffff)) + bf2f((bf16_t)(u.w >> 16)) * bf2f((bf16_t)(v.w >> 16));
      }
    }
    s += __shfl_xor(s, 1, 64);
    if (q < rows && half == 0) delta[q] = s;      // 0 for q >= len
  }
  const int hh = lane >> 5;
  const int kw = wave * 32;                 // this wave's keys
  const bool active = wave < nt;
  // K, V rows of this wave's keys as B operands: X[key = kw + (l&31)][16s + 8hh + j] -- requested BEFORE the wait for the staged
  // images, so that their round trip to memory overlaps the images' instead of following it
  bf16x8 kf[4], vf[4];
  if (active) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = load_frag_global(qbase + HID + (long)(kw + (lane & 31)) * QKV_LD + 16 * s + 8 * hh);
      vf[s] = load_frag_global(qbase + 2 * HID + (long)(kw + (lane & 31)) * QKV_LD + 16 * s + 8 * hh);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
  if (active) {
    const int key = kw + (lane & 31);
    const bool klive = key < len;
    const float madd = p.cu ? (klive ? 0.f : MASK_NEG) : ((p.att_mask && p.att_mask[row0 + key] == 0) ? MASK_NEG : 0.f);
    const float madd2 = madd * LOG2E;          // 0 or -inf
    // dropout_hash2's argument for query 0 of this lane's share (32-bit wrap-around arithmetic, as the element index is defined)
    const uint32_t hbase = (uint32_t)(((long)b * NH + h) * S * S) + (uint32_t)(key & ~1) + (uint32_t)((16 * (lane & 1) + 4 * hh) * S) + p.drop.idx_offset;
    // (recording the forward's dropout decisions as bits and reading them here instead of re-hashing was built and measured:
    // forward +1 us, backward -0.5 us -- the hash hides behind the MFMA / LDS latencies of the loop; not kept)
    for (int qt = 0; qt < ntq; ++qt) {
      f32x16 sa, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sa = mfma32(frag32_row(qimg, qt * 32, s), kf[s], sa);     // S[q][k]
        dp = mfma32(frag32_row(doimg, qt * 32, s), vf[s], dp);    // dP[q][k]
      }
      // dropout decisions: keys 2j and 2j + 1 -- neighbouring lanes -- share one hash per query.  Even lanes hash this lane pair's queries
      // r = 0..7, odd lanes r = 8..15, and the halves are swapped with one DPP move each: 8 hashes per lane instead of 16
      uint32_t hown[8], hoth[8];
      const uint32_t odd = (uint32_t)lane & 1u;
      if constexpr (DROP && CAREL_ATTN_ABLATE != 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // element ((b*NH + h)*S + qh)*S + key with qh = acc32_row(j + 8 * odd, lane) = qt*32 + (j&3) + 8*(j>>2) + 16*odd + 4*hh: the per-lane part
          // is hbase (hoisted), the rest is wave-uniform
          hown[j] = mix32(((hbase + (uint32_t)((qt * 32 + (j & 3) + 8 * (j >> 2)) * S)) >> 1) ^ p.drop.key);
          hoth[j] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hown[j], 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: the neighbour's
        }
      }
      f32x16 pd, dsv;                                             // dropped probabilities, dS (without the 1/sqrt(d): applied to dK / dQ at the end)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // registers 4i..4i+3 of the 32x32 tile are the consecutive queries qt*32 + 8i + 4hh + (0..3): one 16-byte LDS read each
        const f32x4 l4 = *(const f32x4*)(lse + qt * 32 + 8 * i + 4 * hh);
        const f32x4 d4 = *(const f32x4*)(delta + qt * 32 + 8 * i + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * i + e;
          float arg = fmaf(sa[r], 0.125f * LOG2E, madd2 - l4[e]);
          if (REL) arg = fmaf(relb[127 + key - (qt * 32 + 8 * i + 4 * hh + e)], LOG2E, arg);
          const float pr = CAREL_ATTN_ABLATE == 3 ? arg : __builtin_amdgcn_exp2f(arg);
          float dm = 1.0f;
          if constexpr (DROP && CAREL_ATTN_ABLATE != 1) dm = dropout_pick(p.drop, ((uint32_t)(r >> 3) == odd) ? hown[r & 7] : hoth[r & 7], odd);
          pd[r] = pr * dm;
          dsv[r] = pr * fmaf(dp[r], dm, -d4[e]);                  // d loss / d score
        }
      }
      // the bias enters the scores unscaled: its gradient is the plain sum of dS over the diagonal key - query.  Deterministic: each wave
      // adds into its OWN array, and the two half-waves (same keys, queries 4 apart: lane l of the upper half would hit the address
      // of lane l - 4 of the lower) take turns, so no two lanes of one instruction share an address -- plain read-add-write in a
      // fixed order instead of LDS atomics (every other gradient of this library is bit-reproducible).  Dead rows / keys add 0.
      if (REL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* mine = relg + wave * 256 + 127 + key - (qt * 32 + acc32_row(r, lane));
          if (hh == 0) *mine += dsv[r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (hh == 1) *mine += dsv[r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      // dS^T[k][q] -> LDS (4 consecutive q per register group)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = qt * 32 + 8 * i + 4 * hh;
        uint2 v = {pack2bf(dsv[4 * i], dsv[4 * i + 1]), pack2bf(dsv[4 * i + 2], dsv[4 * i + 3])};
        *(uint2*)(dsimg + ds_off(key, q)) = v;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = acc_as_operand(pd, s), df = acc_as_operand(dsv, s);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = mfma32(frag32_tr<true>(doimg, dt * 32, qt * 32 + 16 * s), pf, dv[dt]);   // dV^T[d][k]
          dk[dt] = mfma32(frag32_tr<true>(qimg, dt * 32, qt * 32 + 16 * s), df, dk[dt]);    // dK^T[d][k]
        }
      }
    }
  }
  if (CAREL_ATTN_ABLATE == 2) return;
  __syncthreads();                           // every wave: dO image dead, dS^T complete
  if (REL && threadIdx.x < 255) {                 // the four waves' sums in a fixed order, onto this (sample, head)'s own row
    const float g = ((relg[threadIdx.x] + relg[256 + threadIdx.x]) + relg[512 + threadIdx.x]) + relg[768 + threadIdx.x];
    p.drel[((long)b * NH + h) * 256 + threadIdx.x] += g;
  }
  stage_att(qbase + HID, QKV_LD, rows, doimg);  // K image for the dQ phase
  // dK and dV are stored while the K image is in flight: vmcnt retires in issue order and the image's copies are older than the stores, so
  // a counted wait for all but the 8 store instructions (dense batches: every row is live, the count is static) retires exactly the image.
  // (Packed batches mask rows past the sample -- a fully masked store may be skipped -- and drain everything, as before.)
  char* slot = qimg + wave * 4096;           // the Q image is dead now (every wave has left the main loop): 4 KiB per wave for the row stores
  bf16_t* out = p.dqkv + (row0 + kw) * QKV_LD + h * HD;
  if (active && CAREL_ATTN_ABLATE != 4) {
    store_rows_via_lds(dk, 0.125f, slot, out + HID, QKV_LD, kw, len);       // the 1/sqrt(d) of the scores (a power of two: the same bits as scaling dS)
    store_rows_via_lds(dv, 1.0f, slot, out + 2 * HID, QKV_LD, kw, len);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (active && !p.cu && CAREL_ATTN_ABLATE != 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();              // raw: __syncthreads() would drain the stores too
  __builtin_amdgcn_sched_barrier(0);
  if (!active) return;
  // dQ^T[d][q] = sum_k K^T[d][k] dS^T[k][q]   for this wave's 32 queries q = kw + ..
  f32x16 dq[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
  {
    const int g = lane >> 4, g1 = g & 1, h2 = g >> 1, qq = (lane & 15) >> 2, pp = lane & 3;
    for (int ks = 0; ks < (wave < ntq ? (rows >> 4) : 0); ++ks) {        // (queries past q_rows: dQ = 0, stored below)
      // B operand: dS^T[k = 16ks + 8*h2 + j][q = kw + (l&31)] via transposed reads of the dS^T image
      const int kr0 = 16 * ks + 8 * h2 + qq, kr1 = kr0 + 4;
      const int qcol = kw + 16 * g1 + 4 * pp;
      s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((CAREL_LDS s16x4*)(dsimg + ds_off(kr0, qcol)));
      s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((CAREL_LDS s16x4*)(dsimg + ds_off(kr1, qcol)));
      s16x8 bb = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
      const bf16x8 bf = __builtin_bit_cast(bf16x8, bb);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(frag32_tr<false>(doimg, dt * 32, 16 * ks), bf, dq[dt]);
    }
  }
  if (CAREL_ATTN_ABLATE != 4) store_rows_via_lds(dq, 0.125f, slot, out, QKV_LD, kw, len);
}

}  // namespace carel

using namespace carel;

static int attn_prepare(const carel_attn_args* a, AttnParams* p, const char* who, bool bwd) {
  if (!a) return set_error(CAREL_ERR_ARG, "%s: null args", who);
  if (a->heads != NH || a->head_dim != HD) return set_error(CAREL_ERR_SHAPE, "%s: heads/head_dim must be %d/%d", who, NH, HD);
  if (a->seq_len < 32 || a->seq_len > 128 || (a->seq_len & 31)) return set_error(CAREL_ERR_SHAPE, "%s: seq_len must be 32, 64, 96 or 128 (got %d)", who, a->seq_len);
  if (a->batch <= 0) return set_error(CAREL_ERR_SHAPE, "%s: batch must be positive", who);
  if (!a->qkv || !a->ctx || !a->lse) return set_error(CAREL_ERR_ARG, "%s: null tensor", who);
  if (bwd && (!a->dctx || !a->dqkv)) return set_error(CAREL_ERR_ARG, "%s: null gradient tensor", who);
  p->qkv = (const bf16_t*)a->qkv; p->att_mask = (const long*)a->attention_mask; p->ctx = (bf16_t*)a->ctx;
  p->lse = (float*)a->lse; p->dctx = (const bf16_t*)a->dctx; p->dqkv = (bf16_t*)a->dqkv;
  p->B = a->batch; p->S = a->seq_len; p->cu = (const int*)a->cu_seqlens;
  p->rel = (const float*)a->rel_bias_dist; p->drel = (float*)a->d_rel_bias_dist;
  if (a->q_rows < 0 || a->q_rows > 128 || (a->q_rows & 31)) return set_error(CAREL_ERR_ARG, "%s: q_rows must be 0, 32, 64 or 96", who);
  p->qlim = a->q_rows >= a->seq_len ? 0 : a->q_rows;
  if (bwd && p->rel && !p->drel) return set_error(CAREL_ERR_ARG, "%s: rel_bias_dist needs d_rel_bias_dist in the backward", who);
  p->drop = make_dropout(a->drop_seed, a->drop_site, a->drop_p, a->drop_idx_offset);
  return CAREL_OK;
}

extern "C" int carel_attention_fwd(const carel_attn_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AttnParams p;
  int rc = attn_prepare(a, &p, "carel_attention_fwd", false);
  if (rc) return rc;
  const bool drop = p.drop.thresh != 0;
  if (p.rel) {
    if (drop) hipLaunchKernelGGL((attn_fwd_kernel<true, true>), dim3(p.B * NH), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<true, false>), dim3(p.B * NH), dim3(256), 0, stream, p);
  } else {
    if (drop) hipLaunchKernelGGL((attn_fwd_kernel<false, true>), dim3(p.B * NH), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, false>), dim3(p.B * NH), dim3(256), 0, stream, p);
  }
  return check_launch("attn_fwd_kernel");
}

extern "C" int carel_attention_bwd(const carel_attn_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AttnParams p;
  int rc = attn_prepare(a, &p, "carel_attention_bwd", true);
  if (rc) return rc;
  static bool attr_set = false;     // idempotent; a benign race sets it twice
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_BWD_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_BWD_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_BWD_LDS_REL);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_BWD_LDS_REL);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_attention_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const bool drop = p.drop.thresh != 0;
  if (p.rel) {
    if (drop) hipLaunchKernelGGL((attn_bwd_kernel<true, true>), dim3(p.B * NH), dim3(256), ATTN_BWD_LDS_REL, stream, p);
    else hipLaunchKernelGGL((attn_bwd_kernel<true, false>), dim3(p.B * NH), dim3(256), ATTN_BWD_LDS_REL, stream, p);
  } else {
    if (drop) hipLaunchKernelGGL((attn_bwd_kernel<false, true>), dim3(p.B * NH), dim3(256), ATTN_BWD_LDS, stream, p);
    else hipLaunchKernelGGL((attn_bwd_kernel<false, false>), dim3(p.B * NH), dim3(256), ATTN_BWD_LDS, stream, p);
  }
  return check_launch("attn_bwd_kernel");
}

// ------------------------------------------------------------------------------------------------ MPNet relative positions
// The learned table is relative_attention_bias.weight [32 buckets][12 heads]; bucket[i] (int32 [256], entry i = distance key - query
// = i - 127, entry 255 unused) is computed by the caller with the very expression of transformers
// MPNetEncoder.relative_position_bucket (a float32 log and a truncation: not re-derived here, so no rounding can differ).
namespace carel {
__global__ void relpos_expand_kernel(const float* table, const int* bucket, float* dist) {     // -> dist [NH][256]
  const int h = blockIdx.x, i = threadIdx.x;
  dist[h * 256 + i] = i < 255 ? table[bucket[i] * NH + h] : 0.f;
}
// ddist [batch * NH][256] (one row per (sample, head), see AttnParams.drel) -> dtable [32][NH]: samples in order, distances in order
__global__ __launch_bounds__(256) void relpos_reduce_kernel(const float* ddist, int batch, const int* bucket, float* dtable, int accumulate) {
  __shared__ float bydist[256];
  const int h = blockIdx.x, i = threadIdx.x;
  float s = 0.f;
  for (int b = 0; b < batch; ++b) s += ddist[((long)b * NH + h) * 256 + i];
  bydist[i] = s;
  __syncthreads();
  if (i < 32) {
    float t = 0.f;
    for (int d = 0; d < 255; ++d) if (bucket[d] == i) t += bydist[d];
    dtable[i * NH + h] = accumulate ? dtable[i * NH + h] + t : t;
  }
}
}  // namespace carel

extern "C" int carel_relpos_expand(const void* table, const void* bucket, void* dist, void* stream) {
  if (!table || !bucket || !dist) return set_error(CAREL_ERR_ARG, "carel_relpos_expand: null tensor");
  hipLaunchKernelGGL(relpos_expand_kernel, dim3(NH), dim3(256), 0, (hipStream_t)stream, (const float*)table, (const int*)bucket, (float*)dist);
  return check_launch("relpos_expand_kernel");
}
extern "C" int carel_relpos_reduce(const void* ddist, int32_t batch, const void* bucket, void* dtable, int32_t accumulate, void* stream) {
  if (!ddist || !bucket || !dtable || batch < 1) return set_error(CAREL_ERR_ARG, "carel_relpos_reduce: null tensor or batch < 1");
  hipLaunchKernelGGL(relpos_reduce_kernel, dim3(NH), dim3(256), 0, (hipStream_t)stream, (const float*)ddist, (int)batch, (const int*)bucket, (float*)dtable, (int)accumulate);
  return check_launch("relpos_reduce_kernel");
}
