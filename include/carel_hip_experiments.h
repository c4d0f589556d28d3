/* carel_hip_experiments.h -- entry points that exist ONLY in the experiments build of the library (libcarel_hip_exp.so, compiled with
 * -DCAREL_EXPERIMENTS by carel_vae_amd/build.py): the process-wide tuning hooks the A/B tools and the bitwise-equivalence tests flip, and
 * the kernels that were built, measured and not adopted.  The product library (libcarel_hip.so) exports none of
 * them: every switch is a compile-time constant there (tests/test_abi.py asserts the symbols are absent).
 * Included by carel_hip.h when CAREL_EXPERIMENTS is defined. */
#ifndef CAREL_HIP_EXPERIMENTS_H
#define CAREL_HIP_EXPERIMENTS_H

/* test / tuning hook (process-wide, not thread-safe: set it before any other thread calls the library): 0 = choose the
 * kernel automatically, 1 = 128x128 kernel only, 2 = force the old 256x192 kernel, 3 = the 256 x 96n ping-pong kernel
 * wherever the shape allows; 50+k = the ping-pong kernel takes grids of at least 32*k workgroups (default 192);
 * 30/31 = automatic use of the old 256x192 tile off/on; 20..24 XCD tile layouts of the 128x128 kernel; 70+n = ping-pong tile width
 * 96n forced (0 = heuristic); 90/91 = ping-pong schedule with fine (12-MFMA) / wide (24-MFMA, default) phases; 100+s = weight-gradient
 * split-K factor of the ping-pong kernel forced to s (0 = heuristic); 120/121 = its XCD tile map: row-major chunks / rectangles
 * (default); 130/131 = internal split-K for K >= 1536 only / also for the K = 768 one-row-tile GEMMs (default); 140/141 = the K slices
 * of an internally split NT / NN GEMM on the 128x128 kernel / on the ping-pong kernel where they fit one round (default); 160/161 = the
 * ping-pong kernel's GELU epilogues by erf / exp arithmetic / by table lookup (default; the same bits); 11..19, 61..68 timing
 * ablations (wrong results; only in a -DCAREL_GEMM_ABLATE build).  None of the non-ablation settings changes results beyond the fp32
 * summation order of split-K. */
int carel_gemm_set_variant(int32_t variant);

/* ------------------------------------------------------------------------------------------------
 * Row-band GEMM with the sub-layer tail fused (ABI 5):
 *     h = dropout(A W^T + bias) + resid ;  x = LayerNorm(h) * gamma + beta        for 768-wide outputs
 * Replaces nn.Linear + nn.Dropout + residual add + nn.LayerNorm of HF BertSelfOutput / BertOutput (reached from
 * drl_classifier_ec_mmd_final_mul.py:202-206) in ONE kernel: each workgroup owns 32 complete rows, so the pre-LayerNorm sum never
 * makes a round trip through memory.  Bit-identical to carel_gemm_bf16(CAREL_EPI_BIAS_DROP_RESID) followed by carel_layernorm_fwd.
 * Every workgroup streams the whole weight matrix from L2, so it pays only when M / 32 workgroups fill the chip (M >= ~6000 rows);
 * the encoder uses it for dense batches and keeps the two-kernel path for packed ECPE batches.
 * ---------------------------------------------------------------------------------------------- */
typedef struct carel_gemm_rowln_args {
  const void* A;          /* bf16 [M, K], leading dimension lda (elements, multiple of 8) */
  const void* W;          /* bf16 [768, K] (nn.Linear weight), leading dimension ldb; or, with w_packed = 1, the same matrix in the
                             MFMA-operand order written by carel_gemm_rowln_pack (768 * K elements, ldb ignored): the weight stream is
                             then contiguous per load instruction -- 3-4x the rate of the row-major layout */
  int64_t lda, ldb;
  int32_t M, K;           /* any M >= 1; K multiple of 128 */
  const void* bias;       /* f32 [768] or NULL */
  const void* resid_f32;  /* f32 [M, 768] */
  const void* gamma;      /* f32 [768] */
  const void* beta;       /* f32 [768] */
  float eps;
  void* h_f32;            /* out f32 [M, 768]: the pre-LayerNorm sum (what carel_layernorm_bwd reads), or NULL */
  void* x_f32;            /* out f32 [M, 768] or NULL */
  void* x_bf16;           /* out bf16 [M, 768] or NULL */
  void* stats;            /* out f32 [M, 2] (mean, rstd) or NULL */
  uint32_t drop_seed, drop_site, drop_idx_offset;
  float drop_p;
  const void* drop_row_map; /* optional int32 [M], as in carel_gemm_args */
  int32_t w_packed;       /* 0: W row-major; 1: W packed by carel_gemm_rowln_pack */
} carel_gemm_rowln_args;
int carel_gemm_rowln(const carel_gemm_rowln_args* args, void* stream);
/* out[768 * K] = W[768, K] (leading dimension ldb) re-ordered as [n / 16][k / 64][(k / 32) % 2][(k / 8) % 4][n % 16][k % 8]: what a
 * wave of carel_gemm_rowln loads per instruction is then one contiguous KiB.  Run it whenever the weight changes (after an optimiser step). */
int carel_gemm_rowln_pack(const void* W, int64_t ldb, int32_t K, void* out, void* stream);

#endif
