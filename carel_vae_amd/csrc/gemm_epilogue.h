// Shared pieces of the bf16 MFMA GEMM kernels (gemm.hip: 128x128 tile; gemm_pp.hip: 256 x 96n ping-pong tile):
// the launch parameter block and the fused epilogues.  Replaces nn.Linear forward / backward inside the HF encoder
// layers (drl_classifier_ec_mmd_final_mul.py:202-206, :841).
#pragma once
#include "carel_common.h"
#include "carel_hip_internal.h"

namespace carel {

enum : int {
  EPI_BIAS_BF16 = 0,        // out0(bf16) = acc + bias?            (QKV, dgrad of out-proj)
  EPI_BIAS_GELU = 1,        // out0(bf16) = u = acc+bias ; out1(bf16) = gelu(u)       (FFN1)
  EPI_BIAS_DROP_RESID = 2,  // outf(f32) = dropout(acc + bias) + resid(f32)           (out-proj, FFN2)
  EPI_DGELU_BF16 = 3,       // out0(bf16) = acc * gelu'(aux_bf16)                     (dgrad of FFN2)
  EPI_ADD_F32 = 4,          // outf(f32) = acc + resid(f32)?                          (dgrad of QKV / FFN1)
  EPI_SLAB_F32 = 5,         // outf[z](f32) = acc                                     (wgrad split-K)
  EPI_BIAS_GELU_DG = 6,     // out0(bf16) = gelu'(u) ; out1(bf16) = gelu(u), u = bf16(acc+bias)   (FFN1 when a backward follows)
  EPI_MUL_BF16 = 7,         // out0(bf16) = acc * aux_bf16                             (dgrad of FFN2 on the saved gelu'(u))
};
// The encoder saves gelu'(u) instead of u: the forward epilogue has erf(u / sqrt 2) and exp(-u^2 / 2) in registers anyway, and the
// backward epilogue then costs one multiply per element instead of an erf and an exp (it was VALU-bound on them: ~5 us per tile).
constexpr bool epi_is_gelu(int e) { return e == EPI_BIAS_GELU || e == EPI_BIAS_GELU_DG; }
constexpr bool epi_is_dgelu(int e) { return e == EPI_DGELU_BF16 || e == EPI_MUL_BF16; }      // aux-scaled output + optional column sums
constexpr bool epi_has_bias(int e) { return e == EPI_BIAS_BF16 || epi_is_gelu(e) || e == EPI_BIAS_DROP_RESID; }

struct GemmParams {
  const bf16_t* A; const bf16_t* B;
  long lda, ldb;
  int M, N, K;              // K = contraction length handled by ONE z-slice
  bf16_t* out0; bf16_t* out1; float* outf;
  long ldc;
  const float* bias;        // [N] or null
  const float* resid;       // [M,ldc] f32 or null
  // EPI_BIAS_DROP_RESID only, all three or none: `resid` then holds the PRE-LayerNorm rows h of the LayerNorm whose output is the residual,
  // and the epilogue recomputes LN(h) = (h - mean) * rstd * gamma + beta itself (resid_stats [M][2] = mean, rstd as ln_fwd_kernel stores
  // them; resid_gamma / resid_beta [ldc]) -- the LayerNorm kernel then need not write its f32 output at all (25 MB per sub-layer)
  const float* resid_stats; const float* resid_gamma; const float* resid_beta;
  const bf16_t* aux;        // [M,ldc] bf16 (pre-GELU) for EPI_DGELU
  Dropout drop;
  int tiles_m, tiles_n;
  const int* drop_row_map;  // optional [M]: original row of each packed row (dropout element index)
  float* splitk_ws; size_t splitk_ws_bytes;   // optional workspace enabling the internal split-K path
  float* colsum_a;          // optional, TN form: [splits][M] sums of A over this K-slice (bias gradient)
  float* colsum_part;       // optional [tiles_m][N]: per-row-tile column sums of the epilogue output (bias gradient)
  int split_tile_factor;    // internal split-K heuristic: the caller runs this many equal GEMMs side by side (1 = just this one)
  int xcd_n;                // XCDs laid out as (8/xcd_n) x xcd_n over (M tiles, N tiles); 1 = row-major chunks
  int gelu_lut;             // ping-pong kernel, GELU epilogues: 1 = table lookup (default), 0 = erf / exp arithmetic (tuning hook 160 / 161)
  unsigned* pair_flags;     // ping-pong kernel, pair split-K: [tiles][8 waves] flags in the reserved tail of splitk_ws (zero-initialised once by the caller)
  unsigned pair_seq;        // ... the value that marks THIS launch's partial sums as written (a process-wide launch counter, never 0)
  int epi_prefetch;         // ping-pong kernel: request the epilogue's residual / aux inputs before the main loop (tuning hook 170 / 171)
  int pp_xr, pp_bc;         // ping-pong kernel, NT / NN: the 8 XCDs tile the grid as pp_xr x (8 / pp_xr) rectangles, each walked in
                            // column blocks of pp_bc tiles (so that a round of 32 tiles per XCD is compact); pp_xr = 0: plain chunks
};

// ---------------------------------------------------------------------------------------------
// Epilogue for 4 consecutive columns (col..col+3) of C row `row`, accumulator values v.
// ---------------------------------------------------------------------------------------------
// The inputs of the 4-column epilogue (f32 residual or bf16 aux values of the same 4 elements), so that a kernel can request them
// long before the accumulators are final (gemm_pp.hip asks for a whole tile's inputs before its main loop).
struct EpiIn4 { float4 r; uint2 a; float2 st; };
// LNR = false: this instantiation never recomputes a LayerNorm residual (the wide ping-pong tiles have no registers for its statistics and
// gamma / beta; gemm_pp_pick keeps such GEMMs off them)
template <int EPI, bool LNR = true>
__device__ __forceinline__ void epi_in4(const GemmParams& p, long row, long col, EpiIn4& in) {
  const long off = row * p.ldc + col;
  if (EPI == EPI_BIAS_DROP_RESID || (EPI == EPI_ADD_F32 && p.resid)) in.r = *(const float4*)(p.resid + off);
  if (EPI == EPI_BIAS_DROP_RESID && LNR) { if (p.resid_stats) in.st = *(const float2*)(p.resid_stats + row * 2); }
  if (epi_is_dgelu(EPI)) in.a = *(const uint2*)(p.aux + off);
}
// gamma (ln[0..3]) and beta (ln[4..7]) of the 4 columns, for the recomputed residual (resid_stats); loaded once per column group
template <int EPI>
__device__ __forceinline__ void epi_ln4(const GemmParams& p, long col, float* ln) {
  if (EPI == EPI_BIAS_DROP_RESID) {
    if (p.resid_stats) {
      const float4 g = *(const float4*)(p.resid_gamma + col), b = *(const float4*)(p.resid_beta + col);
      ln[0] = g.x; ln[1] = g.y; ln[2] = g.z; ln[3] = g.w; ln[4] = b.x; ln[5] = b.y; ln[6] = b.z; ln[7] = b.w;
    }
  }
}
template <int EPI, bool LNR = true>
__device__ __forceinline__ void epi_out4(const GemmParams& p, f32x4 v, const EpiIn4& in, long row, long col, const float* ln = nullptr) {
  const long off = row * p.ldc + col;
  if (epi_has_bias(EPI)) {
    if (p.bias) {
      const float4 b = *(const float4*)(p.bias + col);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
  }
  if (EPI == EPI_BIAS_BF16) {
    uint2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    *(uint2*)(p.out0 + off) = o;
  } else if (EPI == EPI_BIAS_GELU) {
    uint2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    if (p.out0) *(uint2*)(p.out0 + off) = o;            // (inference: nobody reads the pre-activation -- out0 may be null)
    // GELU is evaluated on the bf16-rounded pre-activation that backward will read back
    float u0 = bf2f(f2bf(v[0])), u1 = bf2f(f2bf(v[1])), u2 = bf2f(f2bf(v[2])), u3 = bf2f(f2bf(v[3]));
    uint2 g = {pack2bf(gelu_erf(u0), gelu_erf(u1)), pack2bf(gelu_erf(u2), gelu_erf(u3))};
    *(uint2*)(p.out1 + off) = g;
  } else if (EPI == EPI_BIAS_GELU_DG) {
    const f32x2 ua = {bf2f(f2bf(v[0])), bf2f(f2bf(v[1]))}, ub = {bf2f(f2bf(v[2])), bf2f(f2bf(v[3]))};
    f32x2 ga, gb, da, db;
    gelu_erf_both2(ua, ga, da); gelu_erf_both2(ub, gb, db);
    *(uint2*)(p.out0 + off) = uint2{pack2bf(da.x, da.y), pack2bf(db.x, db.y)};
    *(uint2*)(p.out1 + off) = uint2{pack2bf(ga.x, ga.y), pack2bf(gb.x, gb.y)};
  } else if (EPI == EPI_BIAS_DROP_RESID) {
    float4 r = in.r;
    if (LNR && p.resid_stats) {
      const float mean = in.st.x, rstd = in.st.y;
      r.x = ln_apply(r.x, mean, rstd, ln[0], ln[4]); r.y = ln_apply(r.y, mean, rstd, ln[1], ln[5]);
      r.z = ln_apply(r.z, mean, rstd, ln[2], ln[6]); r.w = ln_apply(r.w, mean, rstd, ln[3], ln[7]);
    }
    const uint32_t e = p.drop_row_map ? (uint32_t)((long)p.drop_row_map[row] * p.ldc + col) : (uint32_t)off;
    float4 o;
    float dm[4];
    dropout_mult_n<4>(p.drop, e, dm);
    o.x = v[0] * dm[0] + r.x;
    o.y = v[1] * dm[1] + r.y;
    o.z = v[2] * dm[2] + r.z;
    o.w = v[3] * dm[3] + r.w;
    *(float4*)(p.outf + off) = o;
  } else if (EPI == EPI_DGELU_BF16) {
    const uint2 a = in.a;
    const float u0 = bf2f((bf16_t)(a.x & 0xffff)), u1 = bf2f((bf16_t)(a.x >> 16));
    const float u2 = bf2f((bf16_t)(a.y & 0xffff)), u3 = bf2f((bf16_t)(a.y >> 16));
    uint2 o = {pack2bf(v[0] * gelu_erf_grad(u0), v[1] * gelu_erf_grad(u1)),
               pack2bf(v[2] * gelu_erf_grad(u2), v[3] * gelu_erf_grad(u3))};
    *(uint2*)(p.out0 + off) = o;
  } else if (EPI == EPI_MUL_BF16) {
    const uint2 a = in.a;
    uint2 o = {pack2bf(v[0] * bf2f((bf16_t)(a.x & 0xffff)), v[1] * bf2f((bf16_t)(a.x >> 16))),
               pack2bf(v[2] * bf2f((bf16_t)(a.y & 0xffff)), v[3] * bf2f((bf16_t)(a.y >> 16)))};
    *(uint2*)(p.out0 + off) = o;
  } else if (EPI == EPI_ADD_F32) {
    float4 o = {v[0], v[1], v[2], v[3]};
    if (p.resid) {
      const float4 r = in.r;
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    *(float4*)(p.outf + off) = o;
  } else {  // EPI_SLAB_F32
    float4 o = {v[0], v[1], v[2], v[3]};
    *(float4*)(p.outf + (long)blockIdx.z * p.M * p.ldc + off) = o;
  }
}
// epi_store = the two steps back to back; every kernel's results are those of this one code path
template <int EPI, bool LNR = true>
__device__ __forceinline__ void epi_store(const GemmParams& p, f32x4 v, long row, long col) {
  EpiIn4 in;
  float ln[8];
  if (LNR) epi_ln4<EPI>(p, col, ln);
  epi_in4<EPI, LNR>(p, row, col, in);
  epi_out4<EPI, LNR>(p, v, in, row, col, ln);
}

// ---------------------------------------------------------------------------------------------
// Coalesced epilogue: 8 consecutive columns (col..col+7) of C row `row`; v = fp32 accumulators read back
// from the LDS-staged tile.  8 threads cover 64 columns of one row -> full 128/256-byte lines.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void unpack8(const uint4 a, float* u) {
  u[0] = bf2f((bf16_t)(a.x & 0xffff)); u[1] = bf2f((bf16_t)(a.x >> 16)); u[2] = bf2f((bf16_t)(a.y & 0xffff)); u[3] = bf2f((bf16_t)(a.y >> 16));
  u[4] = bf2f((bf16_t)(a.z & 0xffff)); u[5] = bf2f((bf16_t)(a.z >> 16)); u[6] = bf2f((bf16_t)(a.w & 0xffff)); u[7] = bf2f((bf16_t)(a.w >> 16));
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  return uint4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
}
// The 8-column epilogue in three steps, so that a kernel can issue the loads of the NEXT row block before the stores of
// the current one (a load's s_waitcnt also waits for every older store: loads queued behind stores serialise the store
// round trips).  epi_store8 = the three steps back to back; every kernel's results are those of this one code path.
struct EpiIn8 { float4 r0, r1; uint4 a; float2 st; };
template <int EPI>
__device__ __forceinline__ void epi_bias8(const GemmParams& p, long col, float* b) {
  if (epi_has_bias(EPI) && p.bias) {
    const float4 b0 = *(const float4*)(p.bias + col), b1 = *(const float4*)(p.bias + col + 4);
    b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
  }
}
template <int EPI, bool LNR = true>
__device__ __forceinline__ void epi_in8(const GemmParams& p, long row, long col, EpiIn8& in) {
  const long off = row * p.ldc + col;
  if (EPI == EPI_BIAS_DROP_RESID || (EPI == EPI_ADD_F32 && p.resid)) { in.r0 = *(const float4*)(p.resid + off); in.r1 = *(const float4*)(p.resid + off + 4); }
  if (EPI == EPI_BIAS_DROP_RESID && LNR) { if (p.resid_stats) in.st = *(const float2*)(p.resid_stats + row * 2); }
  if (epi_is_dgelu(EPI)) in.a = *(const uint4*)(p.aux + off);
}
// gamma (ln[0..7]) and beta (ln[8..15]) of the 8 columns, for the recomputed residual (resid_stats); loaded once per column group
template <int EPI>
__device__ __forceinline__ void epi_ln8(const GemmParams& p, long col, float* ln) {
  if (EPI == EPI_BIAS_DROP_RESID) {
    if (p.resid_stats) {
      const float4 g0 = *(const float4*)(p.resid_gamma + col), g1 = *(const float4*)(p.resid_gamma + col + 4);
      const float4 b0 = *(const float4*)(p.resid_beta + col), b1 = *(const float4*)(p.resid_beta + col + 4);
      ln[0] = g0.x; ln[1] = g0.y; ln[2] = g0.z; ln[3] = g0.w; ln[4] = g1.x; ln[5] = g1.y; ln[6] = g1.z; ln[7] = g1.w;
      ln[8] = b0.x; ln[9] = b0.y; ln[10] = b0.z; ln[11] = b0.w; ln[12] = b1.x; ln[13] = b1.y; ln[14] = b1.z; ln[15] = b1.w;
    }
  }
}
// GELU by table (ping-pong kernel): the activation is evaluated on the bf16-ROUNDED pre-activation, so gelu(u) and gelu'(u), both
// rounded to bf16, are functions of 16 bits.  The table holds them (low / high half of a word) for every bf16 u with
// 2^-16 <= |u| < 16 -- 20 exponents x 128 mantissas x 2 signs = 5 120 words, 20 KiB, filled once per process by the same device
// functions the arithmetic path uses, so both paths give the same bits -- and is copied into the LDS behind the staging buffers at
// kernel start.  Finite |u| >= 16 needs no table (gelu = u or -0.0, gelu' = 1 or 0: what the arithmetic gives once exp(-u^2 / 2)
// underflows).  An 8-column group with any lane outside both ranges (|u| < 1.5e-5, Inf, NaN) takes the arithmetic path for the
// whole wave (< 1 % of the groups on N(0, 1) pre-activations).  The erf + exp arithmetic was what bounded this epilogue:
// FFN1 forward 8192 x 3072 x 768: 41-43 us with the bias -> bf16 epilogue, 55-58 with GELU, one output or two
// (tools/bench_ffn1_epilogue.py).
constexpr int LUT_EXP_LO = 111, LUT_NEXP = 20, LUT_HALF = LUT_NEXP * 128, LUT_WORDS = 2 * LUT_HALF, LUT_BYTES = LUT_WORDS * 4;
// Two elements per instruction: the bf16 pair of a 32-bit word is masked, offset (v_pk_add_u16), range-checked (v_pk_min_u16 + one 32-bit
// compare) and turned into two table indices (v_pk_mad_u16 with the sign bits) as a pair; the halves of the two looked-up words are
// merged with one v_perm_b32 per output word.  5.5 VALU instructions per element (the element-by-element form compiled to ~19, and the
// GELU epilogue of the 256 x 192 tile to ~1 900 per lane: 8 us per round of tiles).  A NaN, an Inf or an |u| < 2^-16 anywhere in the wave's
// group sends the group to the arithmetic path (same results); finite |u| >= 16 get their (table-free) words in a pass that only runs
// when the wave holds one.
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
__device__ __forceinline__ bool gelu_lut8(const uint32_t* lut, const uint4 o, uint4& g, uint4& d) {
  const uint32_t w[4] = {o.x, o.y, o.z, o.w};
  u16x2 idx[4], off[4];
  bool ok = true, big_any = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u16x2 a = __builtin_bit_cast(u16x2, w[q] & 0x7fff7fffu);
    const u16x2 lo = {(unsigned short)(LUT_EXP_LO << 7), (unsigned short)(LUT_EXP_LO << 7)};
    const u16x2 i = a - lo;                                                          // wraps past 51 000 below the table
    // finite and not below the table: i < (255 - LUT_EXP_LO) << 7; inside the table: i < LUT_HALF; in between: finite |u| >= 16
    const u16x2 fin = {(unsigned short)(((255 - LUT_EXP_LO) << 7) - 1), (unsigned short)(((255 - LUT_EXP_LO) << 7) - 1)};
    const u16x2 lim = {(unsigned short)(LUT_HALF - 1), (unsigned short)(LUT_HALF - 1)};
    const u16x2 c = __builtin_elementwise_min(i, lim);
    ok = ok && (__builtin_bit_cast(uint32_t, __builtin_elementwise_min(i, fin)) == __builtin_bit_cast(uint32_t, i));
    big_any = big_any || (__builtin_bit_cast(uint32_t, c) != __builtin_bit_cast(uint32_t, i));
    const u16x2 sg = __builtin_bit_cast(u16x2, (w[q] >> 15) & 0x00010001u);
    const u16x2 hf = {(unsigned short)LUT_HALF, (unsigned short)LUT_HALF};
    idx[q] = sg * hf + c;                                                            // < 2 * LUT_HALF always (big values look up the table's last entry)
    off[q] = i;
  }
  if (!__all(ok)) return false;                 // wave-uniform: the whole wave computes this group (|u| < 2^-16, Inf, NaN somewhere)
  uint32_t t[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) { t[2 * q] = lut[idx[q].x]; t[2 * q + 1] = lut[idx[q].y]; }
  if (__any(big_any)) {
    // finite |u| >= 16 needs no table: exp(-u^2 / 2) underflows to 0 and erf is +-1 in fp32, so the arithmetic gives gelu = u / -0.0 and
    // gelu' = 1 / +0.0 -- the word the table would hold
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t h = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
      const uint32_t ie = (e & 1) ? (uint32_t)off[e >> 1].y : (uint32_t)off[e >> 1].x;
      const uint32_t hb = (h & 0x8000u) ? 0x00008000u : (0x3F800000u | h);
      t[e] = ie >= (uint32_t)LUT_HALF ? hb : t[e];
    }
  }
  // word = gelu | gelu' << 16: low halves of a pair -> g, high halves -> d
  g = uint4{__builtin_amdgcn_perm(t[1], t[0], 0x05040100u), __builtin_amdgcn_perm(t[3], t[2], 0x05040100u),
            __builtin_amdgcn_perm(t[5], t[4], 0x05040100u), __builtin_amdgcn_perm(t[7], t[6], 0x05040100u)};
  d = uint4{__builtin_amdgcn_perm(t[1], t[0], 0x07060302u), __builtin_amdgcn_perm(t[3], t[2], 0x07060302u),
            __builtin_amdgcn_perm(t[5], t[4], 0x07060302u), __builtin_amdgcn_perm(t[7], t[6], 0x07060302u)};
  return true;
}

// after the call v[] holds the values that were stored (pre-rounding), for the fused column sums
// lut: the GELU table in LDS (ping-pong kernel) or null (arithmetic)
template <int EPI, bool LNR = true>
__device__ __forceinline__ void epi_out8(const GemmParams& p, float* v, const float* b, const EpiIn8& in, long row, long col, const uint32_t* lut = nullptr,
                                         const float* ln = nullptr) {
  const long off = row * p.ldc + col;
  if (epi_has_bias(EPI)) {
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += b[e];
    }
  }
  if (EPI == EPI_BIAS_BF16) {
    *(uint4*)(p.out0 + off) = pack8(v);
  } else if (EPI == EPI_BIAS_GELU) {
    const uint4 o = pack8(v);
    if (p.out0) *(uint4*)(p.out0 + off) = o;            // block-uniform; null in inference (half the epilogue's bytes)
    uint4 gq, dq;
    if (!(lut && gelu_lut8(lut, o, gq, dq))) {
      float u[8];
      unpack8(o, u);            // GELU of the bf16-rounded pre-activation that backward reads back
#pragma unroll
      for (int e = 0; e < 8; e += 2) { const f32x2 g = gelu_erf2(f32x2{u[e], u[e + 1]}); u[e] = g.x; u[e + 1] = g.y; }
      gq = pack8(u);
    }
    *(uint4*)(p.out1 + off) = gq;
  } else if (EPI == EPI_BIAS_GELU_DG) {
    const uint4 o = pack8(v); // both on the bf16-rounded pre-activation (what the reference's backward would see saved in bf16)
    uint4 gq, dq;
    if (!(lut && gelu_lut8(lut, o, gq, dq))) {
      float u[8], d[8];
      unpack8(o, u);
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        f32x2 g, dg;
        gelu_erf_both2(f32x2{u[e], u[e + 1]}, g, dg);
        u[e] = g.x; u[e + 1] = g.y; d[e] = dg.x; d[e + 1] = dg.y;
      }
      gq = pack8(u); dq = pack8(d);
    }
    *(uint4*)(p.out0 + off) = dq;       // (a streaming / non-temporal store of this backward-only output: measured, no effect -- tools/bench_gemm_chain.py)
    *(uint4*)(p.out1 + off) = gq;
  } else if (EPI == EPI_BIAS_DROP_RESID) {
    float4 r0 = in.r0, r1 = in.r1;
    if (LNR && p.resid_stats) {                // the residual is LN(h): recomputed from the pre-LayerNorm rows (same expression as ln_fwd_kernel)
      const float mean = in.st.x, rstd = in.st.y;
      r0.x = ln_apply(r0.x, mean, rstd, ln[0], ln[8]); r0.y = ln_apply(r0.y, mean, rstd, ln[1], ln[9]);
      r0.z = ln_apply(r0.z, mean, rstd, ln[2], ln[10]); r0.w = ln_apply(r0.w, mean, rstd, ln[3], ln[11]);
      r1.x = ln_apply(r1.x, mean, rstd, ln[4], ln[12]); r1.y = ln_apply(r1.y, mean, rstd, ln[5], ln[13]);
      r1.z = ln_apply(r1.z, mean, rstd, ln[6], ln[14]); r1.w = ln_apply(r1.w, mean, rstd, ln[7], ln[15]);
    }
    const uint32_t e = p.drop_row_map ? (uint32_t)((long)p.drop_row_map[row] * p.ldc + col) : (uint32_t)off;
    float4 o0, o1;
    float dm[8];
    dropout_mult_n<8>(p.drop, e, dm);
    o0.x = v[0] * dm[0] + r0.x; o0.y = v[1] * dm[1] + r0.y;
    o0.z = v[2] * dm[2] + r0.z; o0.w = v[3] * dm[3] + r0.w;
    o1.x = v[4] * dm[4] + r1.x; o1.y = v[5] * dm[5] + r1.y;
    o1.z = v[6] * dm[6] + r1.z; o1.w = v[7] * dm[7] + r1.w;
    *(float4*)(p.outf + off) = o0; *(float4*)(p.outf + off + 4) = o1;
  } else if (EPI == EPI_DGELU_BF16) {
    float u[8];
    unpack8(in.a, u);
#pragma unroll
    for (int e = 0; e < 8; e += 2) { const f32x2 g = gelu_erf_grad2(f32x2{u[e], u[e + 1]}); v[e] *= g.x; v[e + 1] *= g.y; }
    *(uint4*)(p.out0 + off) = pack8(v);
  } else if (EPI == EPI_MUL_BF16) {
    float u[8];
    unpack8(in.a, u);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= u[e];
    *(uint4*)(p.out0 + off) = pack8(v);
  } else if (EPI == EPI_ADD_F32) {
    float4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
    if (p.resid) {
      const float4 r0 = in.r0, r1 = in.r1;
      o0.x += r0.x; o0.y += r0.y; o0.z += r0.z; o0.w += r0.w; o1.x += r1.x; o1.y += r1.y; o1.z += r1.z; o1.w += r1.w;
    }
    *(float4*)(p.outf + off) = o0; *(float4*)(p.outf + off + 4) = o1;
  } else {  // EPI_SLAB_F32
    float* o = p.outf + (long)blockIdx.z * p.M * p.ldc + off;
    *(float4*)o = float4{v[0], v[1], v[2], v[3]}; *(float4*)(o + 4) = float4{v[4], v[5], v[6], v[7]};
  }
}
template <int EPI>
__device__ __forceinline__ void epi_store8(const GemmParams& p, float* v, long row, long col) {
  float b[8], ln[16];
  EpiIn8 in;
  epi_bias8<EPI>(p, col, b);
  epi_ln8<EPI>(p, col, ln);
  epi_in8<EPI>(p, row, col, in);
  epi_out8<EPI>(p, v, b, in, row, col, nullptr, ln);
}

// gemm_rowln.hip: 32-row x 768-column workgroups, GEMM + bias + dropout + residual + LayerNorm in one kernel
struct RowLnParams {
  const bf16_t* A; long lda;          // [M, K] row-major activations
  const bf16_t* W; long ldb;          // [768, K] row-major weight (nn.Linear layout)
  const float* bias;                  // [768]
  const float* resid;                 // [M, 768] f32
  const float* gamma; const float* beta; float eps;
  float* h_out;                       // [M, 768] f32: dropout(A W^T + bias) + resid (what LayerNorm's backward reads); may be null
  float* x_f32; bf16_t* x_bf16;       // LayerNorm output (either may be null)
  float* stats;                       // [M, 2] mean, rstd; may be null
  int M, K;
  Dropout drop; const int* drop_row_map;
};
int gemm_rowln_launch(const RowLnParams& p, bool packed, hipStream_t s);
int gemm_rowln_pack(const void* W, long ldb, int K, void* out, hipStream_t s);
void gemm_rowln_dbg(int d);

// gemm_pp.hip: the 256 x (96 * npn) ping-pong kernel for the NT / NN forms.  gemm_pp_pick returns npn (1..3) if the kernel
// should run this problem, 0 otherwise; force = 1: whenever the shape allows, force = -n: grids of at least n tiles.
int gemm_pp_pick(const GemmParams& p, bool bt, int epi, int force);
int gemm_pp_launch(const GemmParams& p, bool bt, int epi, int npn, hipStream_t s);
#ifdef CAREL_GEMM_ABLATE
int gemm_pp_launch_dbg(const GemmParams& p, int npn, int dbg, hipStream_t s);   // timing ablations (wrong results)
#endif
// weight-gradient form (A^T B into fp32 slabs; p.K = the WHOLE contraction length, dealt to `splits` z slices as evenly
// as possible -- the slices need not be equal, so any split factor works)
int gemm_pp_init_device(int device);     // carel_init: per-device immutable state (the GELU table)
void gemm_pp_force_npn(int n);
void gemm_pp_wide_variant(int on);
void gemm_pp_xcd_rect(int on);
void gemm_pp_gelu_lut(int on);
void gemm_pp_epi_prefetch(int on);
int gemm_pp_pick_tn(const GemmParams& p, int splits);
int gemm_pp_wgrad_splits(int M, int N, long K);
void gemm_pp_wgrad_force(int s);
int gemm_tri_pick(const GemmParams& p, int epi);                 // gemm_tri.hip: the three-group kernel (experiment, hook 220 / 221)
int gemm_tri_launch(const GemmParams& p, int epi, hipStream_t s);
void gemm_tri_enable(int on);
int gemm_pp_launch_tn(const GemmParams& p, int npn, int splits, hipStream_t s);
#ifdef CAREL_GEMM_ABLATE
int gemm_pp_launch_tn_dbg(const GemmParams& p, int npn, int splits, int dbg, hipStream_t s);
#endif
int gemm_pp_launch_slab(const GemmParams& p, bool bt, int npn, int splits, hipStream_t s);
// gemm_sm.hip: 128 x 128 tiles, eight waves, four-stage LDS-DMA ring -- the row-major-A forms at packed row counts (one round of <= 256 workgroups)
int gemm_sm_launch(const GemmParams& p, bool bt, int epi, int splits, hipStream_t s);
// pair split-K (two workgroups per 256 x 192 tile, each half of K; gemm_pp.hip): 1 if this problem should run that way
int gemm_pp_pick_pair(const GemmParams& p, bool bt, int epi);
int gemm_pp_launch_pair(const GemmParams& p, bool bt, int epi, hipStream_t s);
void gemm_pp_pair_enable(int on);
void gemm_pp_group_mode(int m);
constexpr size_t PP_PAIR_FLAG_BYTES = 4096;

}  // namespace carel
