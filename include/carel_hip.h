/* carel_hip.h -- C ABI of libcarel_hip.so, the MI355X (gfx950) implementation of the CAREL-VAE training
 * hot path (reference: drl_classifier_ec_mmd_final_mul.py; "ref:" line numbers below are into that file
 * of tk1363704/CAREL-VAE unless another file is named).
 *
 * The reference has no FFI of its own (it is 100 % Python); the boundary a maintainer binds is the set
 * of torch / transformers calls on the step path.  Each entry point names the reference call it replaces.
 * See INTEGRATION.md for the ctypes stub that goes into the reference's script.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch tensor.data_ptr()) unless marked "host";
 *   - plain-old-data argument structs, no torch types; row-major tensors;
 *   - every call only ENQUEUES work on `stream` (void* = hipStream_t): no device allocation, no host sync.  Process-wide
 *     state the library does keep: the thread-local error string; per device, three library-owned streams and ten events
 *     created on first use (overlap_wgrad / forward chains, see carel_encoder_args); a 20-KiB table of gelu / gelu' by bf16 input in
 *     device memory, filled by carel_init (the one call that synchronises the device), immutable afterwards; the carel_profile_gemm
 *     event log (measurement aid).  None of it changes results.  The product library has NO tuning hooks and no other mutable
 *     process-wide state (ABI 7): carel_gemm_set_variant and the kernels that were built, measured and not adopted live in the
 *     EXPERIMENTS build only (libcarel_hip_exp.so, -DCAREL_EXPERIMENTS; include/carel_hip_experiments.h);
 *   - return 0 on success, a negative CAREL_ERR_* otherwise; carel_last_error() gives the message;
 *   - "bf16" = raw bfloat16 bits (uint16_t); "f32" = IEEE float.
 */
#ifndef CAREL_HIP_H
#define CAREL_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CAREL_OK 0
#define CAREL_ERR_ARG (-1)    /* null pointer / inconsistent arguments */
#define CAREL_ERR_SHAPE (-2)  /* shape not supported by the kernels     */
#define CAREL_ERR_HIP (-3)    /* a HIP runtime call or launch failed    */

/* ABI version of this header; carel_abi_version() must return the same number. */
#define CAREL_ABI_VERSION 7

int carel_abi_version(void);
/* Checks that `device` is a gfx950 part and fills the library's only per-device state, immutable afterwards: the 20-KiB GELU table of
 * the fused FFN1 epilogue (one small kernel + one hipDeviceSynchronize -- the ONLY host synchronisation in the library; no other call
 * synchronises).  Must be called once per device per process before the first carel_gemm_bf16 / carel_encoder_* call on it (those
 * fail with CAREL_ERR_ARG otherwise); calling it again is a no-op.  The host mirror does it when a model is moved to the device
 * (carel_vae_amd._lib.ensure_init).  ref: `model.to(device)` :932 */
int carel_init(int device);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* carel_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Counter-based dropout.  With j = idx + idx_offset and h = mix32((j >> 1) ^ mix32(seed + site*0x9E3779B9)), element j is kept iff
 * the 16-bit half (j & 1) of h is >= floor(p * 2^32) >> 16 (two consecutive elements share one 32-bit hash: the attention kernels take
 * 25 M decisions per layer and direction); kept values are scaled by 1/(1-p).  p <= 0 disables.  `site` numbers the
 * dropout layer (0 embeddings; 1+3l attention probabilities, 2+3l attention-output, 3+3l FFN-output of
 * encoder layer l; 100/101/102 the three classifier-input dropouts, ref :468 :485 :503).
 * Replaces nn.Dropout inside HF BertEmbeddings/BertSelfAttention/BertSelfOutput/BertOutput and
 * `self.dropout` (ref :182).
 * ---------------------------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM with fused epilogues.  Replaces nn.Linear.forward / its backward inside the HF
 * encoder layers (transformers modeling_bert.py BertSelfAttention.query/key/value,
 * BertSelfOutput.dense, BertIntermediate.dense(+GELU), BertOutput.dense; reached from ref :202-206
 * forward and :841 backward).
 * ---------------------------------------------------------------------------------------------- */
#define CAREL_GEMM_NT 0 /* C[M,N] = A[M,K] * B[N,K]^T          forward: x * W^T                 */
#define CAREL_GEMM_NN 1 /* C[M,N] = A[M,K] * B[K,N]            dgrad:   dY * W                  */
#define CAREL_GEMM_TN 2 /* C[M,N] = A[K,M]^T * B[K,N]          wgrad:   dY^T * X  (split-K)     */

#define CAREL_EPI_BIAS_BF16 0       /* out_bf16 = acc (+ bias)                                   */
#define CAREL_EPI_BIAS_GELU 1       /* out_bf16 = u = acc + bias (optional: NULL skips it) ; out2_bf16 = gelu_erf(u) */
#define CAREL_EPI_BIAS_DROP_RESID 2 /* out_f32 = dropout(acc + bias) + resid_f32                 */
#define CAREL_EPI_DGELU_BF16 3      /* out_bf16 = acc * gelu_erf'(aux_bf16)                      */
#define CAREL_EPI_ADD_F32 4         /* out_f32 = acc (+ resid_f32)                               */
#define CAREL_EPI_SLAB_F32 5        /* out_f32[z] = acc of K-slice z   (z < splits)              */
#define CAREL_EPI_BIAS_GELU_DG 6    /* u = bf16(acc + bias): out_bf16 = gelu_erf'(u) ; out2_bf16 = gelu_erf(u)  (NT form; what the
                                       training encoder saves: the backward epilogue then needs no erf / exp)   */
#define CAREL_EPI_MUL_BF16 7        /* out_bf16 = acc * aux_bf16  (NN form; colsum_part as for DGELU)            */

typedef struct carel_gemm_args {
  const void* A;        /* bf16 */
  const void* B;        /* bf16 */
  int64_t lda, ldb, ldc; /* leading dimensions in elements */
  int32_t M, N, K;      /* NT / NN: N multiple of 96 (any M; 256 x 96n kernel) or (M,N) multiples of (128,128) / (256,192);
                           TN: (M,N) multiples of (256,96) or (128,128); K multiple of 64 (of 64*splits for the 128x128 kernel) */
  int32_t form;         /* CAREL_GEMM_* */
  int32_t epilogue;     /* CAREL_EPI_*  */
  int32_t splits;       /* split-K factor (slab epilogue only), else 1 */
  void* out_bf16;
  void* out2_bf16;
  void* out_f32;        /* slab epilogue: [splits][M][ldc] */
  const void* bias;     /* f32 [N] or NULL */
  const void* resid_f32;
  const void* aux_bf16;
  uint32_t drop_seed, drop_site, drop_idx_offset;
  float drop_p;
  const void* drop_row_map; /* optional int32 [M]: original row of each (packed) row for the dropout element index
                               (= row_map[row] * ldc + col); NULL = identity */
  void* splitk_ws;      /* optional f32 workspace: lets small grids (< 256 tiles, K >= 1536) run split-K into slabs
                           followed by one fused-epilogue pass; results equal up to fp32 summation order */
  int64_t splitk_ws_bytes;
  void* colsum_a;       /* optional, CAREL_GEMM_TN only: f32 [splits][M] = sum over the K (token) dimension of A[k][m] per
                           K-slice, i.e. the bias gradient that goes with the weight gradient; computed with one extra
                           ones-vector MFMA per step in the first tile column */
  void* colsum_part;    /* optional, CAREL_EPI_DGELU_BF16 / CAREL_EPI_MUL_BF16 only: f32 [M/128][N] per-row-tile column sums of the output
                           (pre-rounding); summing them over M/128 gives the FFN1 bias gradient */
  int32_t splitk_ws_zeroed; /* (ABI 5) 1 = the caller zero-filled splitk_ws ONCE when it allocated it and lets nobody else write its last
                           4 KiB: enables pair split-K for N = 768 outputs with K >= 1536 at M ~ 8192 (two workgroups per 256 x 192 tile,
                           each half of K; the second waits for the first's partial sums through per-wave flags kept in that tail, one-
                           directionally, so it cannot deadlock).  0 = never (a stale flag in uninitialised memory could end the wait early). */
  /* ABI 6.  CAREL_EPI_BIAS_DROP_RESID only, all three or none: resid_f32 then holds the PRE-LayerNorm rows h of the LayerNorm whose output
   * is the residual, and the epilogue recomputes LN(h) = (h - mean) * rstd * gamma + beta with the expression of carel_layernorm_fwd
   * (bit-identical to reading its f32 output) -- that LayerNorm call may then pass x_f32 = NULL and not write its 4 B/element at all.
   * resid_ln_stats: f32 [M][2] = mean, rstd as carel_layernorm_fwd stores them; resid_ln_gamma / resid_ln_beta: f32 [ldc]. */
  const void* resid_ln_stats; const void* resid_ln_gamma; const void* resid_ln_beta;
} carel_gemm_args;

int carel_gemm_bf16(const carel_gemm_args* args, void* stream);
/* Split-K factor the library wants for the weight gradient dW[M,N] = dY^T X over T tokens (CAREL_GEMM_TN): pass it as
 * `splits` with an out_f32 of [splits][M][ldc] floats, then carel_slab_reduce_f32.  With the 256 x 96n kernel the T/64
 * K tiles are dealt to the slices as evenly as possible, so T need not be a multiple of 64 * splits. */
int32_t carel_gemm_wgrad_splits(int32_t M, int32_t N, int64_t T);
/* Measurement aid (bench.py roofline leg): while enabled, every carel_gemm_bf16 launch is bracketed by
 * HIP events on its stream.  carel_profile_gemm_read() synchronises and returns the summed kernel time
 * (ms), the summed algorithmic flops (2*M*N*K) and the launch count, then resets the log. */
int carel_profile_gemm(int32_t enable, int32_t max_launches);
int carel_profile_gemm_read(double* total_ms, double* total_flops, int64_t* launches);
/* the two calibration medians of the last read-out, us: an event pair around an empty kernel (what _read subtracts from every
 * bracket), and an event pair with nothing in between */
int carel_profile_gemm_overheads(double* empty_kernel_bracket_us, double* event_pair_us);

/* out[n] (+)= sum_z slabs[z][n];  n multiple of 4 */
int carel_slab_reduce_f32(const void* slabs, void* out, int64_t n, int32_t splits, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Grouped weight gradients (ABI 7): up to four dW_g[M_g, N_g] = dY_g^T X_g over the SAME T tokens -- the four nn.Linear weights of an
 * encoder layer: the dW half of `loss.backward()` (drl_classifier_ec_mmd_final_mul.py:841) for BertSelfAttention q/k/v,
 * BertSelfOutput.dense, BertIntermediate.dense, BertOutput.dense -- in ONE launch of the 256 x 96 ping-pong kernel + one small
 * reduction.  The launch's work list holds whole output tiles (full contraction, written straight into dW: no fp32 slabs) for as
 * many rounds of the 256 CUs as the tile count fills, and splits only the remaining tiles into K slices whose compact partial
 * tiles go through `workspace` and are summed in slice order (bit-reproducible).  An encoder layer at T = 8192: 288 tiles = 256 whole +
 * 32 x 8 slices, 25 MB of workspace traffic instead of the 165 MB of slabs that four split-K launches write and read back.
 * No workgroup waits for another one.  db (optional) = column sums of dY = the bias gradient of the same linear, from ones-vector MFMAs
 * in the first tile column.  ln[] (optional): the per-block partials of up to two LayerNorm backward passes (carel_layernorm_bwd's scratch
 * layout), summed into dgamma / dbeta / dbias by the same reduction launch (the arithmetic of carel_layernorm_bwd's own second pass).
 * M_g multiple of 256 (<= 4096), N_g multiple of 96 (<= 3072), T multiple of 64 (>= 256); every pointer 16-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
typedef struct carel_wgrad_problem {
  const void* dY;   /* bf16 [T, M] row-major */
  const void* X;    /* bf16 [T, N] row-major */
  void* dW;         /* f32 [M, N] (nn.Linear weight layout) */
  void* db;         /* f32 [M] or NULL */
  int32_t M, N;
} carel_wgrad_problem;
typedef struct carel_ln_partial_set {
  const void* partials;   /* f32 [carel_layernorm_bwd_blocks(rows)][3 * 768] */
  int64_t rows;
  void* dgamma; void* dbeta; void* dbias;   /* f32 [768] each; any may be NULL */
} carel_ln_partial_set;
typedef struct carel_wgrad_group_args {
  carel_wgrad_problem prob[4];
  int32_t n_prob;         /* 1..4 */
  int64_t T;
  void* workspace;        /* carel_gemm_wgrad_group_ws_bytes(args) bytes (may be 0: every tile whole) */
  int64_t workspace_bytes;
  carel_ln_partial_set ln[2];
  int32_t n_ln;           /* 0..2 */
} carel_wgrad_group_args;
/* workspace size for these shapes (pointers are not read); -1 if the shapes cannot run grouped */
int64_t carel_gemm_wgrad_group_ws_bytes(const carel_wgrad_group_args* args);
int carel_gemm_wgrad_group(const carel_wgrad_group_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embeddings + LayerNorm (HF BertEmbeddings / RobertaEmbeddings.forward) and its backward.
 *   x0 = dropout(LN(word[ids] + pos[position_ids] + type[token_type_ids])), dropout site 0.
 * position_ids: arange(S) for BERT; cumsum(ids != pad) * (ids != pad) + pad for RoBERTa (roberta = 1).
 * ---------------------------------------------------------------------------------------------- */
typedef struct carel_embed_args {
  const void* input_ids;       /* int64 [B, S] */
  const void* token_type_ids;  /* int64 [B, S] or NULL (= 0) */
  const void* word_emb;        /* f32 [vocab, 768]   */
  const void* pos_emb;         /* f32 [max_pos, 768] */
  const void* type_emb;        /* f32 [type_vocab, 768] */
  const void* ln_gamma;        /* f32 [768] */
  const void* ln_beta;         /* f32 [768] */
  float ln_eps;
  int32_t batch, seq_len, hidden;        /* hidden must be 768 */
  int32_t vocab_size, max_pos, type_vocab; /* type_vocab 1 or 2 */
  int32_t roberta, pad_id;
  uint32_t drop_seed, drop_idx_offset;
  float drop_p;
  void* x_f32;                 /* out f32  [B*S, 768] */
  void* x_bf16;                /* out bf16 [B*S, 768] */
  void* stats;                 /* out f32  [B*S, 2] (mean, rstd); input of the backward */
  /* token packing (padding skipped): row t of the outputs is original row tok_row[t] (= b*S + s), -1 = filler
   * row (written as zeros); n_rows rows are produced.  NULL / 0 = dense (row t = original row t, B*S rows). */
  const void* tok_row;         /* int32 [n_rows] or NULL */
  int32_t n_rows;
} carel_embed_args;

int carel_embed_ln_fwd(const carel_embed_args* args, void* stream);
/* number of row blocks of the backward kernels: partial buffers need blocks * slots * 768 floats */
int carel_embed_ln_bwd_blocks(int64_t rows);
/* dword [vocab,768] and dpos [max_pos,768] are ACCUMULATED into with float atomics (zero them first);
 * dtype [type_vocab,768], dgamma, dbeta [768] are overwritten.
 * partials: f32 scratch, carel_embed_ln_bwd_blocks(B*S) * (2 + type_vocab) * 768 floats. */
int carel_embed_ln_bwd(const carel_embed_args* args, const void* dx0_f32, void* dword, void* dpos, void* dtype,
                       void* dgamma, void* dbeta, void* partials, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over 768-wide rows (BertSelfOutput.LayerNorm / BertOutput.LayerNorm) and its backward with
 * the sub-layer dropout backward fused:  dh = LN'(dy) ; dy_bf16 = dh * dropout_mask(site) ;
 * dgamma, dbeta, dbias (= column sum of dy_bf16 before rounding) are overwritten.
 * ---------------------------------------------------------------------------------------------- */
int carel_layernorm_fwd(const void* h_f32, const void* gamma, const void* beta, float eps, int64_t rows, int32_t hidden,
                        void* x_f32, void* x_bf16, void* stats, void* stream);
/* number of partial blocks the backward kernel writes for EXACTLY this row count (4, 8 or 16 rows per block: fewer rows per block at small
 * row counts, so the value is NOT monotonic in `rows` -- 2048 rows: 512 blocks, 2064 rows: 258; size a buffer that is reused for several row
 * counts with the largest value over them; never more than max(512, ceil(rows / 16)) for any smaller count) */
int carel_layernorm_bwd_blocks(int64_t rows);
/* partials: f32 scratch of carel_layernorm_bwd_blocks(rows) * 3 * 768 floats */
int carel_layernorm_bwd(const void* dy_f32, const void* h_f32, const void* stats, const void* gamma, int64_t rows,
                        int32_t hidden, uint32_t drop_seed, uint32_t drop_site, uint32_t drop_idx_offset, float drop_p,
                        void* dh_f32, void* dy_bf16, void* dgamma, void* dbeta, void* dbias, void* partials, void* stream);
/* same with packed rows: drop_row_map int32 [rows] gives each row's original row for the dropout index */
int carel_layernorm_bwd_packed(const void* dy_f32, const void* h_f32, const void* stats, const void* gamma, int64_t rows,
                               int32_t hidden, uint32_t drop_seed, uint32_t drop_site, uint32_t drop_idx_offset, float drop_p,
                               const void* drop_row_map, void* dh_f32, void* dy_bf16, void* dgamma, void* dbeta, void* dbias,
                               void* partials, void* stream);
/* out[c] (+)= sum_p partials[p][c]  (c < n, p < nparts), fixed summation order */
int carel_partial_reduce_f32(const void* partials, void* out, int32_t n, int32_t nparts, int32_t accumulate, void* stream);
/* out[n] (+)= column sums of a bf16 matrix [rows, n] (bias gradients); partials: ceil(rows/256)*n floats */
int carel_colsum_bf16(const void* x_bf16, int64_t ld, int64_t rows, int32_t n, void* out_f32, int32_t accumulate,
                      void* partials, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Self-attention of one encoder layer, all (sample, head) pairs; S in {32,64,96,128}, 12 heads x 64.
 * Replaces transformers BertSelfAttention.forward after the q/k/v projections (eager attention:
 * softmax(QK^T/8 + (1-mask)*finfo.min) -> dropout -> PV) and its backward.
 * qkv / dqkv: bf16 [B*S, 2304] = q | k | v.  ctx / dctx: bf16 [B*S, 768].  lse: f32 [B, 12, S].
 * dropout element index = ((b*12 + h)*S + q)*S + k.
 * ---------------------------------------------------------------------------------------------- */
typedef struct carel_attn_args {
  const void* qkv;
  const void* attention_mask;  /* int64 [B, S], 1 = attend; NULL = all ones */
  void* ctx;                   /* fwd: out; bwd: in */
  void* lse;                   /* fwd: out; bwd: in */
  const void* dctx;            /* bwd in  */
  void* dqkv;                  /* bwd out */
  int32_t batch, seq_len, heads, head_dim;
  uint32_t drop_seed, drop_site, drop_idx_offset;
  float drop_p;
  /* token packing: sample b owns rows [cu_seqlens[b], cu_seqlens[b+1]) of qkv/ctx/dctx/dqkv and attends to exactly
   * those (attention_mask is ignored); seq_len stays the ORIGINAL padded length (dropout index, lse stride). */
  const void* cu_seqlens;      /* int32 [B+1] or NULL */
  /* MPNet relative-position bias (transformers MPNetAttention: attention_scores += position_bias; the encoder of
   * en_ec_sentence_transformer.py:22): f32 [heads][256], entry 127 + (key position - query position), made from the learned
   * [32 buckets][heads] table by carel_relpos_expand; NULL = no bias (BERT / RoBERTa). */
  const void* rel_bias_dist;
  void* d_rel_bias_dist;       /* bwd, required with rel_bias_dist: f32 [batch * heads][256], row (sample, head) ADDED to by that workgroup alone
                                  (no atomics: bit-reproducible); the caller zeroes it once per step -- every layer adds to it -- and folds
                                  it into the table gradient with carel_relpos_reduce */
  int32_t q_rows;              /* ABI 6.  0 = every query row; 32 / 64 / 96: only the first q_rows positions of every sample are queries --
                                  forward: ctx / lse rows past them are NOT written; backward: dctx rows past them are taken as zero and
                                  their dQ rows are written as zeros.  Exact where nothing downstream reads the other rows: the encoder's
                                  last layer when only [CLS] (position 0) is read after it (carel_encoder_args.n_cls) */
} carel_attn_args;

int carel_attention_fwd(const carel_attn_args* args, void* stream);
int carel_attention_bwd(const carel_attn_args* args, void* stream);
/* bucket: int32 [256], bucket[i] = relative_position_bucket(i - 127) (entry 255 unused), computed by the caller */
int carel_relpos_expand(const void* table_f32_32xH, const void* bucket, void* dist_f32_Hx256, void* stream);
int carel_relpos_reduce(const void* ddist_f32_BHx256, int32_t batch, const void* bucket, void* dtable_f32_32xH, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-encoder orchestration over caller-owned buffers.  Replaces `self.encoder(...)` (ref :202-206;
 * transformers BertModel / RobertaModel forward) and the encoder part of `loss.backward()` (ref :841).
 * Weights are read as bf16 (the shadow copy maintained by carel_adam_step / carel_cast_f32_to_bf16),
 * biases / LayerNorm parameters / embeddings as f32.  The q/k/v projections are one fused [2304,768]
 * matrix (query rows, then key, then value).
 *   act     : carel_encoder_act_bytes(B, S, L, inference) bytes; holds every activation the backward
 *             needs (inference = 1: one layer's worth, re-used by every layer; no backward possible)
 *   scratch : carel_encoder_scratch_bytes(B, S) bytes of workspace.  ZERO-FILL IT ONCE when it is allocated, and let nothing but the library
 *             write to it afterwards (ABI 5): a few words of it are flags of the pair split-K GEMMs (carel_gemm_args.splitk_ws_zeroed), which
 *             must not start out equal to a launch sequence number
 *   dx      : f32 [B*S, 768]; in: d(loss)/d(last hidden state); carried down through the layers
 * Constraints: hidden 768, 12 heads, intermediate 3072, seq_len in {32,64,96,128}, B*S % 128 == 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct carel_layer_params {
  const void* qkv_w; const void* qkv_b;     /* bf16 [2304,768], f32 [2304] */
  const void* out_w; const void* out_b;     /* bf16 [768,768],  f32 [768]  */
  const void* ln1_g; const void* ln1_b;     /* f32 [768] */
  const void* ffn1_w; const void* ffn1_b;   /* bf16 [3072,768], f32 [3072] */
  const void* ffn2_w; const void* ffn2_b;   /* bf16 [768,3072], f32 [768]  */
  const void* ln2_g; const void* ln2_b;     /* f32 [768] */
} carel_layer_params;

typedef struct carel_layer_grads {          /* all f32, shapes as above */
  void* qkv_w; void* qkv_b; void* out_w; void* out_b; void* ln1_g; void* ln1_b;
  void* ffn1_w; void* ffn1_b; void* ffn2_w; void* ffn2_b; void* ln2_g; void* ln2_b;
} carel_layer_grads;

typedef struct carel_encoder_args {
  int32_t batch, seq_len, n_layers, hidden, heads, intermediate;
  int32_t vocab_size, max_pos, type_vocab, roberta, pad_id, inference;
  float ln_eps, hidden_dropout, attn_dropout;
  uint32_t drop_seed, drop_row_offset;      /* row offset = global index of this shard's first sample */
  const void* input_ids; const void* attention_mask; const void* token_type_ids;   /* int64 [B,S] */
  const void* word_emb; const void* pos_emb; const void* type_emb; const void* emb_ln_g; const void* emb_ln_b; /* f32 */
  const carel_layer_params* layers;         /* HOST array [n_layers] of device pointers */
  void* act; void* scratch;
  /* token packing (skip padded positions; results identical for prefix-form attention masks):
   * n_tokens = packed row count rounded up to 128, tok_row int32 [n_tokens] (original row or -1),
   * cu_seqlens int32 [B+1].  n_tokens = 0 / NULL pointers = dense. */
  int32_t n_tokens; const void* tok_row; const void* cu_seqlens;
  /* dead-row elimination: nothing but the [CLS] row of the last layer's output is ever read (HF BertPooler), so the last
   * layer runs its row-wise part (out-proj, LayerNorm, FFN, LayerNorm) and their backward on the n_cls (= batch rounded
   * up to 128) [CLS] rows only; results are unchanged.  cls_rows int32 [n_cls]: row of sample i's [CLS] token in the
   * current (dense or packed) row space, -1 for filler; cls_orig_rows int32 [n_cls]: its original row i*S (dropout
   * index), -1 for filler.  The final hidden states (carel_encoder_x_last) are then the n_cls compact rows, and dx is
   * read as n_cls compact rows by the backward of the last layer.  n_cls = 0 disables. */
  int32_t n_cls; const void* cls_rows; const void* cls_orig_rows;
  /* bit 0 (1): carel_encoder_backward_layer enqueues the weight-gradient GEMMs (and their slab / bias-gradient reductions) on a
   * library-owned low-priority second stream, forked by events behind the kernel that produced each dY, so that they
   * run beside the data-gradient chain and the memory-bound LayerNorm / attention backward kernels; bit 1 (2): the
   * forward pass runs the two halves of a dense batch as two chains (second chain on a peer stream).  3 = both.
   * Results are identical (same kernels, same summation order).
   * COMPLETION: in the order of `stream`, after carel_encoder_backward_layer(l) returns the parameter gradients of
   * layer l+1 are complete; those of layer l after the next call, carel_encoder_backward_join or
   * carel_encoder_backward_embeddings.  0: everything on `stream`, every layer complete when its call returns. */
  int32_t overlap_wgrad;
  const carel_layer_grads* layer_grads;     /* HOST array [n_layers] */
  void* d_word_emb; void* d_pos_emb; void* d_type_emb; void* d_emb_ln_g; void* d_emb_ln_b;
  void* dx;
  /* MPNet relative-position bias, shared by all layers (see carel_attn_args): NULL = none.  d_rel_bias_dist is accumulated into by
   * every layer's attention backward; the caller zeroes it before the first carel_encoder_backward_layer of a step. */
  const void* rel_bias_dist; void* d_rel_bias_dist;
} carel_encoder_args;

/* The library's two low-priority streams of the current device (created on first use, never destroyed):
 * 0 = the weight-gradient stream used by overlap_wgrad, 1 = an auxiliary stream the host side uses for work that may
 * trail the backward pass (the per-layer Adam updates of FusedAdam(fuse_into_backward=True)).  NULL on failure. */
void* carel_side_stream(int32_t which);
int carel_encoder_backward_join(const carel_encoder_args* args, void* stream);
int64_t carel_encoder_act_bytes(int32_t batch, int32_t seq_len, int32_t n_layers, int32_t inference);
int64_t carel_encoder_scratch_bytes(int32_t batch, int32_t seq_len);
/* device pointer (inside act) of the final hidden states, f32 [B*S, 768] */
void* carel_encoder_x_last(const carel_encoder_args* args);
int carel_encoder_forward(const carel_encoder_args* args, void* stream);
/* fp32 DEBUG forward (forward only, dropout off, dense batches): the same graph with every stored value in fp32 -- linears on the
 * f32-input matrix cores (carel_sgemm_f32), attention, GELU and LayerNorm in fp32 -- so that a difference from the fp32 reference
 * (BertModel.forward, drl_classifier_ec_mmd_final_mul.py:202-206) is kernel error, not bf16 rounding (tests assert <= 1e-5 against
 * the reference's own fp32 outputs).  Here args->layers[i].*_w point to F32 weights (same shapes); args->act / scratch / n_cls are
 * not used.  work: carel_encoder_f32_work_bytes(batch, seq_len) bytes; x_out: f32 [batch*seq_len, 768] final hidden states.
 * seq_len <= 128.  ~20x slower than carel_encoder_forward: a measurement tool, not a product path. */
int64_t carel_encoder_f32_work_bytes(int32_t batch, int32_t seq_len);
int carel_encoder_forward_f32(const carel_encoder_args* args, void* work, void* x_out_f32, void* stream);
/* One layer of the backward pass.  Call for layer = n_layers-1 down to 0 after a training forward: a pass STARTS with the last layer
 * (with token packing that call also clears the filler rows of the shared dqkv scratch, which the later calls of the pass rely on). */
int carel_encoder_backward_layer(const carel_encoder_args* args, int32_t layer, void* stream);
int carel_encoder_backward_embeddings(const carel_encoder_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * VAE tail of DrlClassifier.forward (ref :202-261): pooler -> latent heads -> sample -> emotion CE,
 * cause BCE, pair BCE-with-logits(pos_weight), RBF-MMD, annealed KL, decoder softmax + BCE; and the
 * backward of all of it.  fp32 throughout.
 *
 *   carel_tail_latents : pooled = tanh(W_p x_last[:,0] + b_p) (HF BertPooler); lat = [mu_e|lv_e|mu_c|lv_c]
 *                        (ref :312-336).  Also the front half of get_pair_preds (ref :266-275).
 *   carel_tail_losses  : z, terms[0..8] and d(loss)/d(parameter) for the classifier heads + decoder,
 *                        d(loss)/d lat kept in `work`.   terms = {partial, mmd, emo, cau, pair, kl_e,
 *                        kl_c, rec, loss}; loss = w_mmd*(-mmd) + w_emo*emo + w_cau*cau + w_pair*pair +
 *                        kl_e + kl_c + rec (ref :256-261); kl_* already multiplied by kl_weight.
 *   carel_tail_backward: latent heads + pooler backward -> d_head_*, d_pooler_*, dx_last (zero except
 *                        the CLS rows), all scaled by *grad_out_dev (NULL = 1).
 *   carel_pair_probs   : sigmoid(pair_classifier([mu_e + eps_e e^lv_e, mu_c + eps_c e^lv_c])) (ref :277-282)
 * Data-parallel hooks (all optional): global_label_sum/global_n give the pos_weight of the GLOBAL batch;
 * z_global [global_n, 2*ec_dim] (all-gathered samples) + global_row_offset make the MMD the global-batch
 * statistic, differentiated for the local rows and scaled by mmd_grad_scale (= world size when gradients
 * are averaged over ranks).
 * ---------------------------------------------------------------------------------------------- */
typedef struct carel_tail_args {
  int32_t batch, seq_len, hidden, ec_dim, e_classes, bow_dim;
  const void* x_last_f32;            /* f32 [B*S, 768] last encoder LayerNorm output */
  const void* pooler_w; const void* pooler_b;
  const void* head_w[4]; const void* head_b[4];   /* emotion_mu, emotion_log_var, cause_mu, cause_log_var */
  const void* emo_w; const void* emo_b;
  const void* cau_w; const void* cau_b;
  const void* pair_w; const void* pair_b;
  const void* dec_w; const void* dec_b;
  const void* emo_labels;            /* int64 [B] */
  const void* cau_labels;            /* f32 [B] */
  const void* pair_labels;           /* f32 [B] */
  const void* bow;                   /* f32 [B, bow_dim] */
  const void* eps_e; const void* eps_c; /* f32 [ec_dim] */
  float w_mmd, w_emo, w_cau, w_pair, kl_weight, label_smoothing;
  float drop_p; uint32_t drop_seed, drop_row_offset;
  float mmd_alpha, mmd_eps;
  int32_t dis_mode;                  /* disentanglement term: 0 = -w_mmd * RBF-MMD (ref :231-233, :256);
                                        1 = +w_mmd * HSIC(z_e, z_c) (drl_classifier_ec_hsic.py:214, :253; w_mmd = 1 there); 2 = none */
  int32_t emo_bce;                   /* 1 = one-logit sigmoid + BCE emotion head of the ablation scripts
                                        (drl_classifier_ec_hsic.py:455-470; e_classes must be 1, labels 0/1), 0 = softmax CE (ref :461-476) */
  const void* global_label_sum;      /* f32 [1] or NULL */
  int32_t global_n, global_row_offset;
  const void* z_global;              /* f32 [global_n, 2*ec_dim] or NULL */
  float mmd_grad_scale;
  int32_t global_rank_stride;        /* 0: z_global is dense; else floats between the blocks of `batch` rows contributed by
                                        consecutive ranks (lets one all-gather carry z plus a few extra floats per rank) */
  int32_t global_label_ranks;        /* > 1: the global label sum is the sum of this many floats starting at
                                        global_label_sum, global_rank_stride apart (each rank's own sum, straight out of
                                        the same all-gather); 0 / 1: global_label_sum[0] is already the total */
  /* outputs */
  void* pooled;                      /* f32 [B, 768] */
  void* lat;                         /* f32 [B, 4*ec_dim] */
  void* z;                           /* f32 [B, 2*ec_dim] */
  void* terms;                       /* f32 [16] */
  void* work;                        /* f32 [carel_tail_workspace_floats(...)] */
  void* d_emo_w; void* d_emo_b; void* d_cau_w; void* d_cau_b; void* d_pair_w; void* d_pair_b;
  void* d_dec_w; void* d_dec_b;
  void* d_head_w[4]; void* d_head_b[4];   /* may be NULL (heads are not optimised, ref :292-295) */
  void* d_pooler_w; void* d_pooler_b;
  void* dx_last_f32;                 /* f32 [B*S, 768] (packed: [n_rows, 768]) */
  const void* cls_rows;              /* int32 [B]: row of each sample's [CLS] token in x_last / dx_last; NULL = b*seq_len */
  int32_t n_rows;                    /* rows of dx_last to clear (0 = B*seq_len) */
  int32_t serial;                    /* ABI 6.  0: carel_tail_losses may run its single-workgroup loss kernel on the library's side stream
                                        (carel_side_stream(0)) beside the decoder passes on `stream`, forked and joined with events inside
                                        the call -- `stream` order is all a caller ever sees; 1: every kernel on `stream`, in order
                                        (serial kernel traces, per-launch timing) */
} carel_tail_args;

int64_t carel_tail_workspace_floats(int32_t batch, int32_t ec_dim, int32_t bow_dim);
/* pooled, lat and -- when eps_e / eps_c / z are given -- the sampled embeddings z (so that a data-parallel caller can
 * all-gather z before carel_tail_losses, which recomputes the same z) */
int carel_tail_latents(const carel_tail_args* args, void* stream);
int carel_tail_losses(const carel_tail_args* args, void* stream);
/* Measurement aid: while a device buffer of 16 int64 is registered, the fused loss kernel of carel_tail_losses writes
 * 100 MHz time stamps at its phase boundaries into it (NULL switches it off). */
int carel_tail_profile(void* dev_i64_x16);
int carel_tail_backward(const carel_tail_args* args, const void* grad_out_dev_f32, void* stream);
/* same, plus an additional upstream gradient on the sampled embeddings z (f32 [B, 2*ec_dim], NOT scaled by grad_out):
 * lets further loss terms defined on z_e / z_c (e.g. the CLUB bound of the VI ablation) reach the encoder */
int carel_tail_backward_dz(const carel_tail_args* args, const void* grad_out_dev_f32, const void* dz_extra_f32, void* stream);
/* Host->device feeding of the bag-of-words targets (`bow_reps.to(device)`, ref :829, 6 MB per batch when dense): the batch's
 * non-zeros as triples -- trip = int32 rows[nnz], int32 cols[nnz], f32 vals[nnz], contiguous, distinct (row, col) -- are
 * expanded into the dense f32 [B, V] block the loss kernels read (zero fill + scatter, stream-ordered). */
int carel_bow_expand(const void* trip, int32_t nnz, void* out_f32, int32_t B, int32_t V, void* stream);
/* HOST function (no HIP call, no GIL when reached through ctypes): gather the dataset rows idx[0..batch) into one staging
 * block of 4-byte words -- the seven tensors of a batch (ref :136-144, :823-830) back to back at the given word offsets
 * (int64 fields 8-byte aligned), the bag-of-words targets as rows[B*M], cols[B*M] (-1 = padding), vals[B*M] at off_trip.
 * Every source pointer is a HOST array stacked over the n_samples of the dataset. */
typedef struct carel_host_pack_args {
  const void* input_ids; const void* attention_masks; const void* token_type_ids;   /* int64 [n, S] */
  const void* labels; const void* cau_labels;                                     /* f32 [n] */
  const void* emo_labels;                                                         /* int64 [n] or f32 [n] (emo_is_float) */
  const void* bow_cols; const void* bow_vals;                                     /* int32 / f32 [n, M] */
  const int64_t* idx;                                                             /* [batch] sample indices */
  void* dst;                                                                      /* staging block (page-locked) */
  int64_t n_samples;
  int32_t batch, seq_len, bow_entries, emo_is_float;
  int64_t off_input_ids, off_attention_masks, off_token_type_ids, off_labels, off_cau_labels, off_emo_labels, off_trip;
  /* optional token packing (padding skipped, carel_encoder_args.tok_row / cu_seqlens): lengths = int32 [n] attended
   * length of every sample; writes cu_seqlens int32 [batch_padded + 1] at off_cu and tok_row int32 [<= batch*seq_len rounded up to 128] at
   * off_tok, returns the attended-token count and its round-up to 128 in t_eff / t_pad.  lengths = NULL: skipped. */
  const void* lengths;
  int32_t batch_padded;
  int64_t off_cu, off_tok;
  int64_t t_eff, t_pad;                                                           /* out */
} carel_host_pack_args;
int carel_host_pack_batch(carel_host_pack_args* args);
/* x[i] *= *scale_dev  (device scalar; used to apply loss.backward()'s grad_output without a host sync) */
int carel_scale_f32(void* x_f32, int64_t n, const void* scale_dev_f32, void* stream);
/* offset (in floats, inside `work`) of the flag carel_tail_losses sets to 1.0 when the pair loss was
 * replaced by 0 (ref :510-511); carel_adam_step's skip_flag points at it */
int64_t carel_tail_pair_dead_offset(int32_t batch, int32_t ec_dim, int32_t bow_dim);
int carel_pair_probs(const void* lat, const void* eps_e, const void* eps_c, const void* pair_w, const void* pair_b,
                     int32_t batch, int32_t ec_dim, void* prob, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused Adam over a flat parameter buffer.  Replaces torch.optim.Adam(...).step() (ref :936, :842) with
 * torch's defaults and update order (lerp for exp_avg; sqrt(v)/sqrt(bc2) + eps).  Optionally refreshes
 * the bf16 shadow copy the GEMMs read.  [skip_lo, skip_hi) is left untouched when *skip_flag != 0
 * (the pair head when its loss term was replaced by 0, ref :510-511, has grad None in the reference).
 * ---------------------------------------------------------------------------------------------- */
typedef struct carel_adam_args {
  void* param; const void* grad; void* exp_avg; void* exp_avg_sq;   /* f32 [n], 16-byte aligned */
  void* shadow_bf16;                                                 /* bf16 [n] or NULL */
  int64_t n;
  int64_t step;                       /* 1-based step count (bias correction) */
  float lr, beta1, beta2, eps;
  float grad_scale;                   /* multiplies the gradient first (0 = 1.0) */
  int64_t skip_lo, skip_hi; const void* skip_flag;
  /* AdamW / clipping (the sentence-transformer fine-tune, fit(): AdamW weight_decay 0.01 on weight matrices only,
   * max_grad_norm 1): all optional, zero / NULL = plain Adam */
  const void* grad_scale_dev;         /* device f32 scalar multiplied into the gradient as well (carel_grad_norm_clip's coefficient) */
  float weight_decay;                 /* decoupled: param *= 1 - lr * weight_decay before the update, inside decay_segments */
  const void* decay_segments;         /* device int64 [n_decay_segments][2]: sorted [start, end) element ranges relative to
                                         `param` (multiples of 4) */
  int32_t n_decay_segments;
  /* optional device f32 scalar owned by the caller (zero it once): how many steps [skip_lo, skip_hi) has been frozen so far.
   * The range's bias corrections then use step - *skip_count (torch.optim.Adam advances a parameter's step counter only
   * when it has a gradient), and the call adds 1 to it when *skip_flag != 0. */
  void* skip_count;
} carel_adam_args;
int carel_adam_step(const carel_adam_args* args, void* stream);
int carel_cast_f32_to_bf16(const void* src_f32, void* dst_bf16, int64_t n, void* stream);
/* torch.optim.RMSprop(params, lr).step() with torch's defaults (alpha 0.99, eps 1e-8, no momentum, not centered): the
 * optimiser of the five discriminators of drl_classifier_en.py (:1056-1060).  fp32 only (the discriminators have no bf16 copy). */
int carel_rmsprop_step(void* param_f32, const void* grad_f32, void* square_avg_f32, int64_t n, float lr, float alpha, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sentence-embedding fine-tuning (chi_ec_sentence_transformer.py / en_ec_sentence_transformer.py :22, :78, :84-87; the
 * arithmetic is in the third-party `sentence_transformers`, absent here: parity unpinned, oracle/carel_oracle_st.py restates
 * the published algorithm).  fp32.
 *   carel_mean_pool_*: Pooling(mode "mean") over the attended tokens of the last hidden states x f32 [rows, hidden]:
 *     sample b owns rows row0[b] .. row0[b] + len[b] (prefix-form masks); backward writes EVERY row of dx (zeros outside).
 *   carel_triplet_semihard: losses.BatchSemiHardTripletLoss(model, margin), Euclidean distance, forward and (demb != NULL)
 *     the gradient w.r.t. the embeddings for upstream gradient 1; batch <= 64; labels int32.
 *   carel_grad_norm_clip: out2 = {||grad||_2, min(1, max_norm / (norm + 1e-6))} (torch.nn.utils.clip_grad_norm_);
 *     scratch = 1024 floats; pass out2 + 1 as carel_adam_args.grad_scale_dev.
 *   carel_l2_normalize_*: models.Normalize (the third module of all-mpnet-base-v2): y = x / max(||x||_2, 1e-12) per row; norm f32
 *     [rows] is saved for the backward dx = (g - y (y . g)) / norm.
 * ---------------------------------------------------------------------------------------------- */
int carel_mean_pool_fwd(const void* x_f32, const void* row0_i32, const void* len_i32, int32_t batch, int32_t hidden, void* out_f32, void* stream);
int carel_mean_pool_bwd(const void* g_f32, const void* row_sample_i32, const void* len_i32, int64_t rows, int32_t hidden, void* dx_f32, void* stream);
int carel_triplet_semihard(const void* emb_f32, const void* labels_i32, int32_t batch, int32_t hidden, float margin, void* loss_out_f32,
                           void* demb_f32, void* stream);
int carel_grad_norm_clip(const void* grad_f32, int64_t n, float max_norm, void* scratch_f32, void* out2_f32, void* stream);
int carel_l2_normalize_fwd(const void* x_f32, int32_t rows, int32_t hidden, void* y_f32, void* norm_f32, void* stream);
int carel_l2_normalize_bwd(const void* g_f32, const void* y_f32, const void* norm_f32, int32_t rows, int32_t hidden, void* dx_f32, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RBF-MMD statistic.  Replaces MMDStatistic.__call__ (ref :547-569) + pdist (ref :580-589) and
 * their autograd backward.  mmd = 2*a01*sum(K12) + a00*(sum(K11)-tr K11) + a11*(sum(K22)-tr K22),
 * K = sum_alpha exp(-alpha * (eps + |d2|)), a00 = 1/(n1(n1-1)), a11 = 1/(n2(n2-1)), a01 = -1/(n1 n2).
 * ---------------------------------------------------------------------------------------------- */
typedef struct carel_mmd_args {
  const void* s1;       /* f32 [n1, d], row stride ld1 */
  const void* s2;       /* f32 [n2, d], row stride ld2 */
  int64_t ld1, ld2;
  int32_t n1, n2, d;    /* d <= 64 */
  int32_t n_alphas;     /* 1..8 */
  float alphas[8];
  float eps;            /* 1e-5 in the reference */
  void* mmd_out;        /* f32 [1] */
  void* kernels_out;    /* optional f32 [(n1+n2)^2]  (ret_matrix=True) or NULL */
  /* backward only: */
  const void* grad_mmd; /* f32 [1] upstream gradient */
  void* g1;             /* f32 [n1, d] contiguous: d mmd / d s1 * grad */
  void* g2;             /* f32 [n2, d] contiguous */
} carel_mmd_args;

int carel_rbf_mmd_fwd(const carel_mmd_args* args, void* stream);
int carel_rbf_mmd_bwd(const carel_mmd_args* args, void* stream);

/* Replaces `pdist(sample_1, sample_2, norm=2, eps)` (ref :580-589, the live L2 branch) and its autograd backward:
 * dist[i][j] = sqrt(eps + |  |s1_i|^2 + |s2_j|^2 - 2 s1_i . s2_j  |), computed from the squared distance itself (no
 * exp / log round trip), so far-apart samples (d2 >> 100) and near-coincident ones are both exact to fp32 rounding. */
typedef struct carel_pdist_args {
  const void* s1;       /* f32 [n1, d], row stride ld1 */
  const void* s2;       /* f32 [n2, d], row stride ld2 */
  int64_t ld1, ld2;
  int32_t n1, n2, d;    /* d <= 64 */
  float eps;            /* 1e-5 in the reference */
  void* dist_out;       /* fwd: f32 [n1, n2] */
  const void* grad_dist;/* bwd: f32 [n1, n2] upstream gradient */
  void* g1;             /* bwd: f32 [n1, d] contiguous */
  void* g2;             /* bwd: f32 [n2, d] contiguous */
} carel_pdist_args;
int carel_pdist_fwd(const carel_pdist_args* args, void* stream);
int carel_pdist_bwd(const carel_pdist_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HSIC statistic (ablation head).  Replaces HSIC / GaussianKernelMatrix / pairwise_distances of
 * drl_classifier_ec_hsic.py:529-547 and their backward: tr(L H K H)/(m-1)^2, K = exp(-D(x)/s_x),
 * L = exp(-D(y)/s_y), D = squared Euclidean distances, H = I - 11^T/m.
 * ---------------------------------------------------------------------------------------------- */
typedef struct carel_hsic_args {
  const void* x; const void* y;     /* f32 [m, d], row strides ldx / ldy */
  int64_t ldx, ldy;
  int32_t m, d;                     /* d <= 64 */
  float s_x, s_y;                   /* 1 in the reference */
  void* hsic_out;                   /* f32 [1] */
  const void* grad_hsic;            /* bwd: f32 [1] upstream gradient (NULL = 1) */
  void* gx; void* gy;               /* bwd: f32 [m, d] contiguous */
} carel_hsic_args;
int carel_hsic_fwd(const carel_hsic_args* args, void* stream);
int carel_hsic_bwd(const carel_hsic_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * VI / CLUB head (ablation script drl_classifier_ec_vi.py).  z = [e | c] f32 [B, 2*ec_dim] are the sampled emotion /
 * cause embeddings; net[8] = approximation network p(e|c): ec_mu.0.weight, ec_mu.0.bias, ec_mu.2.weight, ec_mu.2.bias,
 * ec_log_var.0.weight, .0.bias, .2.weight, .2.bias (:156-163).
 *   carel_vi_aprx : get_ec_aprx_loss (:422-427) on c.detach() -> loss_out and d_net[8] (gradients of the net only)
 *   carel_vi_upper: get_ec_upper_loss (:429-440) with negatives e[perm] -> loss_out and dz = d loss / d z [B, 2*ec_dim]
 *                   (feed it to carel_tail_backward's dz_extra, scaled by beta)
 * ---------------------------------------------------------------------------------------------- */
typedef struct carel_vi_args {
  const void* z;
  int32_t batch, ec_dim;             /* ec_dim <= 32 */
  const void* net[8];
  const void* perm;                  /* int32 [B], upper only */
  void* loss_out;                    /* f32 [1] */
  void* d_net[8];                    /* aprx only */
  void* dz;                          /* upper only, f32 [B, 2*ec_dim] (overwritten) */
} carel_vi_args;
int carel_vi_aprx(const carel_vi_args* args, void* stream);
int carel_vi_upper(const carel_vi_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tail of the three-space adversarial model of drl_classifier_en.py (config 4): replaces everything after
 * `pooler_output` in DrlClassifier.forward (drl_classifier_en.py:220-334) and the gradient side of the six
 * backward calls of its training step (:919-939).  fp32 throughout.
 *
 * Latent layout  lat [B, 2*con_dim + 4*ec_dim] = content_mu | content_log_var | emotion_mu | emotion_log_var |
 *                                                cause_mu | cause_log_var               (:227-232, :378-415)
 * Sample layout  z   [B, 2*ec_dim + con_dim]   = emotion | cause | content  == generative_emb (:243);
 *                eps  [2*ec_dim + con_dim] in the same layout (the reference draws content, emotion, cause in that
 *                order, :238-240 -- the caller fills eps accordingly).
 * terms [32]: 0..6 = content_disc_loss_emo, content_disc_loss_cau, emotion_disc_loss, ec_disc_loss,
 *             cause_disc_loss, ce_disc_loss, vae_and_classifier_loss (the tuple forward returns, :334);
 *             7,8 content entropies (emo, cau); 9 emotion_disc entropy; 10 cause_disc entropy; 11 ec_disc entropy;
 *             12 ce_disc entropy; 13 emo_mul; 14 cau_mul; 15 content_mul; 16 pair_mul; 17 kl_e; 18 kl_c;
 *             19 kl_content (the three already multiplied by their annealed weights); 20 reconstruction.
 *
 * Every discriminator reads a DETACHED embedding (:425-515), so its loss and its entropy term reach only its own
 * parameters.  carel_en_tail_losses writes, for unit upstream gradients:
 *   g_cdisc_*[0], [1] : d content_disc_loss_emo / d content_disc, d content_disc_loss_cau / d content_disc
 *   g_cdisc_*[2]      : d vae_loss / d content_disc   (= con_adv_weight * both entropies)
 *   g_sdisc_*[i]      : d (own discriminator loss) / d {emotion_disc, cause_disc, ec_disc, ce_disc}[i]
 *   g_sdisc_ent_*[i]  : d vae_loss / d the same (weighted entropy)
 *   d_*               : d vae_loss / d {content_classifier, emotion/cause/pair classifier, decoder}
 * and keeps d vae_loss / d lat in `work`; carel_en_tail_backward turns that into d_pooler_* and dx_last, all
 * scaled by *grad_out_dev.  The six latent heads are not in any optimiser group (:357-376): no gradient is
 * produced for them.  Dropout masks: counter-based, sites 110..119 in the order of the ten nn.Dropout calls.
 * ---------------------------------------------------------------------------------------------- */
typedef struct carel_en_tail_args {
  int32_t batch, seq_len, hidden, ec_dim, con_dim, bow_dim;
  const void* x_last_f32;            /* f32 [rows, 768] last encoder LayerNorm output */
  const void* cls_rows;              /* int32 [B] or NULL (= b*seq_len) */
  int32_t n_rows;                    /* rows of dx_last to clear (0 = B*seq_len) */
  const void* pooler_w; const void* pooler_b;
  const void* head_w[6]; const void* head_b[6];     /* content_mu, content_log_var, emotion_mu, emotion_log_var, cause_mu, cause_log_var */
  const void* cdisc_w; const void* cdisc_b;         /* content_disc [V, ec_dim] */
  const void* sdisc_w[4]; const void* sdisc_b[4];   /* emotion_disc [1,con_dim], cause_disc [1,con_dim], ec_disc [1,ec_dim], ce_disc [1,ec_dim] */
  const void* ccls_w; const void* ccls_b;           /* content_classifier [V, con_dim] */
  const void* emo_w; const void* emo_b; const void* cau_w; const void* cau_b;   /* [1, ec_dim] */
  const void* pair_w; const void* pair_b;           /* [1, 2*ec_dim] */
  const void* dec_w; const void* dec_b;             /* [V, 2*ec_dim + con_dim] */
  const void* emo_labels; const void* cau_labels; const void* pair_labels;     /* f32 [B] */
  const void* bow;                                  /* f32 [B, V] */
  const void* eps;                                  /* f32 [2*ec_dim + con_dim] */
  float w_con_adv, w_ec_adv, w_ecce_adv, w_ec_mul, w_con_mul, w_pair;           /* :325-330 */
  float kl_w_ec, kl_w_con;                          /* annealed KL weights (1 once iteration >= kl_ann_iterations) */
  float label_smoothing, epsilon;
  float drop_p; uint32_t drop_seed;
  uint32_t drop_row_offset;                         /* data parallel: first GLOBAL sample index of this shard (dropout masks hash
                                                       the global element index) */
  const void* global_label_sum;                     /* f32 [1] or NULL: sum of pair labels over the GLOBAL batch (pos_weight, :599) */
  int32_t global_n;                                 /* global batch size (with global_label_sum) */
  /* outputs */
  void* pooled;                      /* f32 [B, 768] */
  void* lat;                         /* f32 [B, 2*con_dim + 4*ec_dim] */
  void* z;                           /* f32 [B, 2*ec_dim + con_dim] */
  void* terms;                       /* f32 [32] */
  void* work;                        /* f32 [carel_en_tail_workspace_floats(...)] */
  void* g_cdisc_w[3]; void* g_cdisc_b[3];
  void* g_sdisc_w[4]; void* g_sdisc_b[4];
  void* g_sdisc_ent_w[4]; void* g_sdisc_ent_b[4];
  void* d_ccls_w; void* d_ccls_b; void* d_emo_w; void* d_emo_b; void* d_cau_w; void* d_cau_b;
  void* d_pair_w; void* d_pair_b; void* d_dec_w; void* d_dec_b;
  void* d_pooler_w; void* d_pooler_b;
  void* dx_last_f32;                 /* f32 [n_rows, 768] */
} carel_en_tail_args;
int64_t carel_en_tail_workspace_floats(int32_t batch, int32_t ec_dim, int32_t con_dim, int32_t bow_dim);
int carel_en_tail_latents(const carel_en_tail_args* args, void* stream);     /* pooled, lat (also the front of get_pair_preds, :336-349) */
int carel_en_tail_losses(const carel_en_tail_args* args, void* stream);      /* z, terms and every gradient listed above */
int carel_en_tail_backward(const carel_en_tail_args* args, const void* grad_out_dev_f32, void* stream);
/* get_pair_preds (:336-353): raw pair logits from lat with fresh emotion / cause noise (f32 [ec_dim] each) */
int carel_en_pair_logits(const void* lat, int32_t lat_stride, int32_t emo_off, int32_t cau_off, const void* eps_e, const void* eps_c,
                         const void* pair_w, const void* pair_b, int32_t batch, int32_t ec_dim, void* logits, void* stream);
/* dst = (accumulate ? dst : 0) + *scale_dev * src   (scale_dev NULL = 1): how a discriminator's .grad takes one
 * backward call's share without a host sync */
int carel_axpy_f32(void* dst_f32, const void* src_f32, int64_t n, const void* scale_dev_f32, int32_t accumulate, void* stream);
/* fp32 GEMM used by the vocabulary-wide heads (exposed for the parity tests): C[M,N] (+)= op(A) op(B)
 * ta = 0: A is [M,K] row-major (lda); 1: A is [K,M].  tb = 0: B is [N,K] row-major (ldb); 1: B is [K,N].
 * bias [N] or NULL.  splits > 1: C receives `splits` partial results c_split_stride floats apart (no bias). */
int carel_sgemm_f32(const void* A, int64_t lda, int32_t ta, const void* B, int64_t ldb, int32_t tb, void* C, int64_t ldc,
                    int32_t M, int32_t N, int32_t K, const void* bias, int32_t accumulate, int32_t splits, int64_t c_split_stride,
                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * Hardware-layout self test (MFMA fragment maps, transposed LDS reads, LDS-DMA staging) used by
 * tests/test_gpu_layouts.py: raw dumps of what the helpers produce on exact integer data (layout of
 * the two buffers is documented in csrc/selftest.hip); the test compares them with numpy.
 * ---------------------------------------------------------------------------------------------- */
int carel_selftest_layouts(const void* in_bf16_40960, void* out_f32_73728, void* stream);

#ifdef CAREL_EXPERIMENTS
#include "carel_hip_experiments.h"
#endif
#ifdef __cplusplus
}
#endif
#endif /* CAREL_HIP_H */
