// Internal host-side helpers shared by the translation units of libcarel_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/carel_hip.h"
#include "carel_common.h"

// Tuning hooks exist in the EXPERIMENTS build only (-DCAREL_EXPERIMENTS -> libcarel_hip_exp.so: carel_gemm_set_variant, the
// kernels that were built, measured and not adopted -- gemm_tri.hip, gemm_rowln.hip, pair split-K -- and every switch the A/B tools
// flip).  In the product library each of these is a compile-time constant: no mutable process-wide state, dead branches folded away.
#ifdef CAREL_EXPERIMENTS
#define CAREL_TUNABLE(type, name, value) static type name = value
#else
#define CAREL_TUNABLE(type, name, value) static constexpr type name = value
#endif

namespace carel {

int set_error(int code, const char* fmt, ...);
int check_launch(const char* what);
int slab_reduce_multi(const void* slabs, void* out, int64_t n, const void* slabs2, void* out2, int64_t n2, int splits, const void* partials,
                      int nparts, void* dgamma, void* dbeta, void* dbias, hipStream_t stream);   // gemm.hip: weight slabs + bias partials + LayerNorm partials in one launch
#ifdef CAREL_EXPERIMENTS
int gemm_rowln_wanted(long rows);         // gemm.hip: should a 768-wide linear + LayerNorm of this many rows run as the fused row-band kernel?
int gemm_rowln_wanted_k(int K);           // ... also for this contraction length (tuning hook: the K = 3072 form can be switched off alone)
#else
inline int gemm_rowln_wanted(long) { return 0; }      // (the row-band kernel exists in the experiments build only)
inline int gemm_rowln_wanted_k(int) { return 0; }
#endif
void encoder_ln_resid_enable(int on);      // encoder.hip: LayerNorm residuals recomputed by the next epilogue (tuning hook 230 / 231)
void encoder_ln_slab_fusion_enable(int on); // encoder.hip: split-K slab epilogues fused into the following LayerNorm (tuning hook 270 / 271)
void encoder_wgrad_group_enable(int on);    // encoder.hip: one grouped weight-gradient launch per layer (tuning hook 240 / 241)
void adam_grid_cap(long cap);            // adam.hip: workgroups per Adam launch (tuning hook 280 + k)
void tail_overlap_enable(int on);          // tail.hip: loss kernel beside the decoder passes (tuning hook 210 / 211)
int gemm_pp_init_device(int device);      // gemm_pp.hip: fills the GELU table (carel_init)
// 768-wide row gather / scatter by int32 index (ln.hip); either of the f32 / bf16 pairs may be null
int gather_rows(const void* in_f32, const void* in_bf16, const void* idx, int n, void* out_f32, void* out_bf16, hipStream_t stream);
int scatter_rows(const void* in_f32, const void* in_bf16, const void* idx, int n, void* out_f32, void* out_bf16, hipStream_t stream);

// LayerNorm backward in two launches that may go to different streams (ln.hip; see carel_layernorm_bwd_packed)
int layernorm_bwd_rows(const void* dy, const void* h, const void* stats, const void* gamma, int64_t rows, uint32_t drop_seed, uint32_t drop_site,
                       uint32_t drop_idx_offset, float drop_p, const void* drop_row_map, void* dh_f32, void* dy_bf16, void* partials,
                       hipStream_t stream);
int layernorm_bwd_reduce(const void* partials, int64_t rows, void* dgamma, void* dbeta, void* dbias, hipStream_t stream);
int layernorm_bwd_blocks_max(int64_t rows);    // the most partial blocks carel_layernorm_bwd_blocks(r) returns for any r <= rows (scratch sizing)
// the same row kernel with its input gradient rows taken as  sum_z slabs[z][row] (+ resid[row])  -- the deferred epilogue (CAREL_EPI_ADD_F32) of a
// split-K data-gradient GEMM, with that epilogue's order of additions (bit-identical to slab epilogue + layernorm_bwd_rows)
int layernorm_bwd_rows_slabs(const void* slabs, int splits, const void* resid, const void* h, const void* stats, const void* gamma, int64_t rows,
                             uint32_t drop_seed, uint32_t drop_site, uint32_t drop_idx_offset, float drop_p, const void* drop_row_map, void* dh_f32,
                             void* dy_bf16, void* partials, hipStream_t stream);
// ln.hip: the slab epilogue of a split-K out-projection / FFN2 (bias + dropout + residual, stored or recomputed LN(resid)) and the LayerNorm
// behind it in one launch (packed batches; GEMM_EX_DEFER_EPILOGUE leaves the slabs in the workspace)
int layernorm_fwd_slabs(const void* slabs, int splits, const void* bias, const void* resid, const void* resid_stats, const void* resid_gamma,
                        const void* resid_beta, uint32_t drop_seed, uint32_t drop_site, uint32_t drop_idx_offset, float drop_p, const void* drop_row_map,
                        void* h_out, const void* gamma, const void* beta, float eps, int64_t rows, void* x_f32, void* x_bf16, void* stats,
                        hipStream_t stream);

// carel_gemm_bf16 with the split-K heuristic told that `split_tile_factor` equal GEMMs run side by side (the forward's
// half-batch chains): the split factor is then chosen as for ONE GEMM over all their rows -- same K partition, same bits; gemm.hip
// (| GEMM_EX_FIXED_ROWS: M is the same whatever the batch -- the [CLS]-only last layer -- so a finer K partition is allowed)
constexpr int GEMM_EX_FIXED_ROWS = 0x100;
// (round 4) | GEMM_EX_DEFER_EPILOGUE: if this GEMM runs split-K into slabs, do NOT launch its slab epilogue: the caller's next kernel reads
// the slabs itself (LayerNorm fused behind the split GEMMs of packed ECPE batches: layernorm_bwd_rows_slabs / layernorm_fwd_slabs).  Only
// together with a plan: gemm_bf16_split_plan() tells how many slabs the same call would write (1 = it does not split: do not defer).
constexpr int GEMM_EX_DEFER_EPILOGUE = 0x200;
constexpr int GEMM_EX_PLAN_ONLY = 0x400;       // (internal to gemm.hip)
int gemm_bf16_ex(const carel_gemm_args* a, int split_tile_factor, void* stream);
int gemm_bf16_split_plan(const carel_gemm_args* a, int split_tile_factor);       // slabs the internal split-K path would write for this call; 1 = single pass
// largest value carel_gemm_wgrad_splits(M, N, T) can take under any tuning-hook setting (slab buffer sizing); gemm.hip
int gemm_wgrad_splits_max(int M, int N, long T);
int embed_ln_bwd_ex(const carel_embed_args* a, const void* dx0, void* dword, void* dpos, void* dtype_, void* dgamma, void* dbeta,
                    void* partials, void* row_scratch, hipStream_t stream, const void* sort_ws = nullptr);
// ln.hip: the embedding tables' gradients from sorted (id, row) keys instead of float atomics (deterministic); the keys are made in the forward pass
int embed_sort_supported(const carel_embed_args* a);
size_t embed_sort_bytes();
int embed_ln_fwd_keys(const carel_embed_args* a, void* sort_ws, hipStream_t stream);
int embed_sort_rows(const carel_embed_args* a, void* sort_ws, hipStream_t stream);

// grouped weight gradients (gemm_pp.hip, round 4): up to four dW[M, N] = dY^T X over the same T tokens in ONE launch + one small reduction
struct WgradGroupProb { const void* dY; const void* X; void* dW; void* db; int M, N; };      // dY bf16 [T, M], X bf16 [T, N], dW f32 [M, N], db f32 [M] or null
struct WgradGroupLn { const void* partials; int nparts; void* dgamma; void* dbeta; void* dbias; };   // LayerNorm-backward partials [nparts][3 * 768] summed by the same reduction
size_t gemm_pp_wgrad_group_ws_bytes(const WgradGroupProb* pb, int n, long T);
int gemm_pp_wgrad_group_ok(const WgradGroupProb* pb, int n, long T);
int gemm_pp_wgrad_group(const WgradGroupProb* pb, int n, long T, void* ws, size_t ws_bytes, const WgradGroupLn* ln, int n_ln, hipStream_t stream);   // NT / NN, `splits` K slices -> fp32 slabs at p.outf

inline Dropout make_dropout(uint32_t seed, uint32_t site, float p, uint32_t idx_offset) {
  Dropout d;
  d.key = mix32(seed + site * 0x9E3779B9u);
  d.idx_offset = idx_offset;
  if (p <= 0.f) { d.thresh = 0u; d.scale = 1.f; }
  else if (p >= 1.f) { d.thresh = 0xFFFFFFFFu; d.scale = 0.f; }
  else {
    d.thresh = (uint32_t)((double)p * 4294967296.0);
    d.scale = (float)(1.0 / (1.0 - (double)p));
    if (d.thresh == 0u) d.scale = 1.f;
  }
  return d;
}

}  // namespace carel
