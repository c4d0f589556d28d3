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
 *   - every call only ENQUEUES work on `stream` (void* = hipStream_t): no allocation, no host sync, no
 *     global mutable state besides the thread-local error string;
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
#define CAREL_ABI_VERSION 1

int carel_abi_version(void);
/* Checks that `device` is a gfx950 part and records nothing else.  ref: `model.to(device)` :932 */
int carel_init(int device);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* carel_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Counter-based dropout.  keep(element) = mix32((idx + idx_offset) ^ mix32(seed + site*0x9E3779B9))
 * >= floor(p * 2^32); kept values are scaled by 1/(1-p).  p <= 0 disables.  `site` numbers the
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
#define CAREL_EPI_BIAS_GELU 1       /* out_bf16 = u = acc + bias ; out2_bf16 = gelu_erf(u)       */
#define CAREL_EPI_BIAS_DROP_RESID 2 /* out_f32 = dropout(acc + bias) + resid_f32                 */
#define CAREL_EPI_DGELU_BF16 3      /* out_bf16 = acc * gelu_erf'(aux_bf16)                      */
#define CAREL_EPI_ADD_F32 4         /* out_f32 = acc (+ resid_f32)                               */
#define CAREL_EPI_SLAB_F32 5        /* out_f32[z] = acc of K-slice z   (z < splits)              */

typedef struct carel_gemm_args {
  const void* A;        /* bf16 */
  const void* B;        /* bf16 */
  int64_t lda, ldb, ldc; /* leading dimensions in elements */
  int32_t M, N, K;      /* M,N multiples of 128; K multiple of 64*splits */
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
} carel_gemm_args;

int carel_gemm_bf16(const carel_gemm_args* args, void* stream);
/* out[n] (+)= sum_z slabs[z][n];  n multiple of 4 */
int carel_slab_reduce_f32(const void* slabs, void* out, int64_t n, int32_t splits, int32_t accumulate, void* stream);

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

/* ------------------------------------------------------------------------------------------------
 * Hardware-layout self test (MFMA fragment maps, transposed LDS reads, LDS-DMA staging) used by
 * tests/test_gpu_layouts.py: raw dumps of what the helpers produce on exact integer data (layout of
 * the two buffers is documented in csrc/selftest.hip); the test compares them with numpy.
 * ---------------------------------------------------------------------------------------------- */
int carel_selftest_layouts(const void* in_bf16_40960, void* out_f32_73728, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAREL_HIP_H */
