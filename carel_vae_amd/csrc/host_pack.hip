// Pure HOST code of libcarel_hip.so (no HIP call, no device pointer): compiled by hipcc into the library, and -- by itself, with
// -DCAREL_HOST_ONLY -- by g++ -fsanitize=address,undefined for tests/test_host_pack_sanitized.py (sanitizers on the host C path;
// GPU sanitizers are not available on this pool).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/carel_hip.h"

#ifdef CAREL_HOST_ONLY
namespace carel {
static thread_local char g_host_err[512] = "";
static int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_host_err, sizeof(g_host_err), fmt, ap);
  va_end(ap);
  return code;
}
}
extern "C" const char* carel_last_error(void) { return carel::g_host_err; }
#else
namespace carel { int set_error(int code, const char* fmt, ...); }
#endif
using namespace carel;

// ---------------------------------------------------------------------------------------------------------------------
// Host-side batch assembly for the input pipeline (carel_vae_amd.data.PrefetchLoader): gathers the rows `idx` of the
// dataset's stacked arrays straight into one page-locked staging block, in the layout the device side unpacks (the seven
// `batch[k].to(device)` tensors of ref :823-830 as one block; bag-of-words as padded entry lists).  Pure host code: called
// through ctypes, i.e. WITHOUT the Python GIL, from the loader's background thread -- a Python-level gather there would
// fight the training thread for the interpreter lock.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int carel_host_pack_batch(carel_host_pack_args* a) {
  if (!a || !a->idx || !a->dst || !a->input_ids || !a->attention_masks || !a->token_type_ids || !a->labels || !a->cau_labels || !a->emo_labels ||
      !a->bow_cols || !a->bow_vals)
    return set_error(CAREL_ERR_ARG, "carel_host_pack_batch: null pointer");
  if (a->batch < 1 || a->seq_len < 1 || a->bow_entries < 1 || a->n_samples < 1) return set_error(CAREL_ERR_ARG, "carel_host_pack_batch: bad sizes");
  const int B = a->batch, S = a->seq_len, M = a->bow_entries;
  int32_t* w = (int32_t*)a->dst;
  int64_t* ids = (int64_t*)(w + a->off_input_ids);
  int64_t* att = (int64_t*)(w + a->off_attention_masks);
  int64_t* tt = (int64_t*)(w + a->off_token_type_ids);
  float* lab = (float*)(w + a->off_labels);
  float* cau = (float*)(w + a->off_cau_labels);
  int32_t* rows = w + a->off_trip;
  int32_t* cols = rows + (size_t)B * M;
  float* vals = (float*)(cols + (size_t)B * M);
  for (int b = 0; b < B; ++b) {
    const int64_t i = a->idx[b];
    if (i < 0 || i >= a->n_samples) return set_error(CAREL_ERR_ARG, "carel_host_pack_batch: sample index %lld out of range", (long long)i);
    memcpy(ids + (size_t)b * S, (const int64_t*)a->input_ids + (size_t)i * S, (size_t)S * 8);
    memcpy(att + (size_t)b * S, (const int64_t*)a->attention_masks + (size_t)i * S, (size_t)S * 8);
    memcpy(tt + (size_t)b * S, (const int64_t*)a->token_type_ids + (size_t)i * S, (size_t)S * 8);
    lab[b] = ((const float*)a->labels)[i];
    cau[b] = ((const float*)a->cau_labels)[i];
    if (a->emo_is_float) ((float*)(w + a->off_emo_labels))[b] = ((const float*)a->emo_labels)[i];
    else ((int64_t*)(w + a->off_emo_labels))[b] = ((const int64_t*)a->emo_labels)[i];
    for (int m = 0; m < M; ++m) rows[(size_t)b * M + m] = b;
    memcpy(cols + (size_t)b * M, (const int32_t*)a->bow_cols + (size_t)i * M, (size_t)M * 4);
    memcpy(vals + (size_t)b * M, (const float*)a->bow_vals + (size_t)i * M, (size_t)M * 4);
  }
  a->t_eff = 0; a->t_pad = 0;
  if (a->lengths) {
    // token packing of the batch (DrlClassifier._pack_info): cu[b] = first packed row of sample b (samples past B repeat
    // the total), tok[t] = original row b*S + s of packed row t, -1 for the filler rows up to the next multiple of 128
    if (a->batch_padded < B) return set_error(CAREL_ERR_ARG, "carel_host_pack_batch: batch_padded < batch");
    int32_t* cu = w + a->off_cu;
    int32_t* tok = w + a->off_tok;
    long t = 0;
    for (int b = 0; b < B; ++b) {
      long len = ((const int32_t*)a->lengths)[a->idx[b]];
      if (len < 0) len = 0;
      if (len > S) len = S;
      cu[b] = (int32_t)t;
      for (long sidx = 0; sidx < len; ++sidx) tok[t + sidx] = (int32_t)((long)b * S + sidx);
      t += len;
    }
    for (int b = B; b <= a->batch_padded; ++b) cu[b] = (int32_t)t;
    const long tp = (t + 127) / 128 * 128;
    for (long x = t; x < tp; ++x) tok[x] = -1;                  // the region holds roundup128(B * S) entries
    a->t_eff = t; a->t_pad = tp;
  }
  return CAREL_OK;
}
