// Host-side orchestration of the BERT-base encoder forward / backward (12 x {QKV GEMM, attention,
// out-proj GEMM + dropout + residual, LayerNorm, FFN1 GEMM + GELU, FFN2 GEMM + dropout + residual,
// LayerNorm}) over caller-owned activation and scratch buffers.  Replaces
// `self.encoder(input_ids, attention_mask, token_type_ids)` (drl_classifier_ec_mmd_final_mul.py:202-206,
// transformers BertModel.forward) and the encoder part of `loss.backward()` (:841).
// Nothing is allocated here; every call only enqueues kernels on the caller's stream.
#include <cstdlib>
#include "carel_hip_internal.h"

using namespace carel;

namespace {

constexpr int EH = 768, EI = 3072, ENH = 12;

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct LayerAct {
  char* xin_bf16; char* qkv; char* lse; char* ctx; char* h1; char* st1; char* x1_bf16; char* u; char* g; char* h2; char* st2;
};
struct ActLayout {
  size_t per_layer, total;
  size_t o_xin, o_qkv, o_lse, o_ctx, o_h1, o_st1, o_x1, o_u, o_g, o_h2, o_st2;
  size_t o_embst, o_xa, o_xb, o_cctx, o_cxres, o_layers;
};

ActLayout act_layout(long B, long S, int L, int inference) {
  const size_t T = (size_t)B * S;
  ActLayout a; size_t o = 0;
  a.o_xin = o; o += al(T * EH * 2);
  a.o_qkv = o; o += al(T * 3 * EH * 2);
  a.o_lse = o; o += al((size_t)B * ENH * S * 4);
  a.o_ctx = o; o += al(T * EH * 2);
  a.o_h1 = o; o += al(T * EH * 4);
  a.o_st1 = o; o += al(T * 2 * 4);
  a.o_x1 = o; o += al(T * EH * 2);
  a.o_u = o; o += al(T * EI * 2);
  a.o_g = o; o += al(T * EI * 2);
  a.o_h2 = o; o += al(T * EH * 4);
  a.o_st2 = o; o += al(T * 2 * 4);
  a.per_layer = o;
  size_t g = 0;
  a.o_embst = g; g += al(T * 2 * 4);
  a.o_xa = g; g += al(T * EH * 4);
  a.o_xb = g; g += al(T * EH * 4);
  const size_t Bc = ((size_t)B + 127) / 128 * 128;          // compact [CLS] rows of the last layer
  a.o_cctx = g; g += al(Bc * EH * 2);
  a.o_cxres = g; g += al(Bc * EH * 4);
  a.o_layers = g;
  a.total = g + a.per_layer * (inference ? 1 : (size_t)L);
  return a;
}

LayerAct layer_act(const ActLayout& a, char* base, int l, int inference) {
  char* p = base + a.o_layers + (inference ? 0 : (size_t)l * a.per_layer);
  LayerAct r;
  r.xin_bf16 = p + a.o_xin; r.qkv = p + a.o_qkv; r.lse = p + a.o_lse; r.ctx = p + a.o_ctx; r.h1 = p + a.o_h1; r.st1 = p + a.o_st1;
  r.x1_bf16 = p + a.o_x1; r.u = p + a.o_u; r.g = p + a.o_g; r.h2 = p + a.o_h2; r.st2 = p + a.o_st2;
  return r;
}

struct Scratch { char* dy; char* dyb; char* dyb2; char* du; char* dctx; char* dqkv; char* slabs; char* ws; char* part; char* part2; char* part3; };
// The operands of a layer's weight gradients -- dyb, dyb2, du, dqkv and the two LayerNorm-backward partial buffers -- exist TWICE, used by
// layers of even / odd index: the layer's weight gradients run as one grouped launch on the side stream after its attention backward
// and may still be reading them while the main stream is already writing the next layer's (round 4; 113 MB at T = 8192).
struct ScratchLayout { size_t o_dy, o_dyb[2], o_dyb2[2], o_du[2], o_dctx, o_dqkv[2], o_slabs, o_ws, o_part[2], o_part2, o_part3[2], o_sort, ws_bytes, slab_bytes, total; };

// split-K factor of the weight-gradient GEMMs (K = tokens): chosen by the GEMM library for the kernel it will run
int wgrad_splits(long T, int M, int N) { return carel_gemm_wgrad_splits(M, N, T); }

ScratchLayout scratch_layout(long B, long S) {
  const size_t T = (size_t)B * S;
  ScratchLayout s; size_t o = 0;
  s.o_dy = o; o += al(T * EH * 4);
  for (int par = 0; par < 2; ++par) {
    s.o_dyb[par] = o; o += al(T * EH * 2);
    s.o_dyb2[par] = o; o += al(T * EH * 2);      // LN1-backward output, so that the FFN2 weight gradient may still read dyb (side stream)
    s.o_du[par] = o; o += al(T * EI * 2);
    s.o_dqkv[par] = o; o += al(T * 3 * EH * 2);
  }
  s.o_dctx = o; o += al(T * EH * 2);
  size_t slab = 0;
  const int shapes[4][2] = {{EH, EI}, {EI, EH}, {EH, EH}, {3 * EH, EH}};
  for (auto& sh : shapes) {     // sized for the largest split count either GEMM kernel may choose (the tuning hooks can switch kernels later)
    const size_t n = (size_t)gemm_wgrad_splits_max(sh[0], sh[1], (long)T) * ((size_t)sh[0] * sh[1] + sh[0]) * 4;
    slab = n > slab ? n : slab;
  }
  {   // ... and for the split tiles of the grouped launch
    const WgradGroupProb pb[4] = {{nullptr, nullptr, nullptr, nullptr, EH, EI}, {nullptr, nullptr, nullptr, nullptr, EI, EH}, {nullptr, nullptr, nullptr, nullptr, 3 * EH, EH},
                                  {nullptr, nullptr, nullptr, nullptr, EH, EH}};
    const size_t n = gemm_pp_wgrad_group_ws_bytes(pb, 4, (long)T);
    slab = n > slab ? n : slab;
  }
  s.o_slabs = o; o += al(slab); s.slab_bytes = al(slab);      // weight-gradient slabs (side stream when overlapping)
  s.o_ws = o; o += al(slab); s.ws_bytes = al(slab);      // split-K workspace of the forward / data-gradient GEMMs (main stream)
  size_t part = (size_t)layernorm_bwd_blocks_max((long)T) * 4 * EH * 4;       // (a packed batch of fewer rows may use MORE, smaller blocks: ln.hip)
  const size_t cs = ((T + 255) / 256) * EI * 4;
  part = part > cs ? part : cs;
  for (int par = 0; par < 2; ++par) {
    s.o_part[par] = o; o += al(part);       // LN2-backward partials (and the embedding backward's)
    s.o_part3[par] = o; o += al(part);      // LN1-backward partials          } their reductions run on the side stream while the main stream moves on
  }
  s.o_part2 = o; o += al(part);             // DGELU column-sum partials
  s.o_sort = o; o += al(embed_sort_bytes());  // (token id, position id) of every row + their sorted keys: the embedding tables' gradients without atomics
  s.total = o;
  return s;
}

Scratch scratch_of(const ScratchLayout& l, char* b, int par = 0) {
  Scratch s; s.dy = b + l.o_dy; s.dyb = b + l.o_dyb[par]; s.dyb2 = b + l.o_dyb2[par]; s.du = b + l.o_du[par]; s.dctx = b + l.o_dctx; s.dqkv = b + l.o_dqkv[par];
  s.slabs = b + l.o_slabs; s.ws = b + l.o_ws; s.part = b + l.o_part[par]; s.part2 = b + l.o_part2; s.part3 = b + l.o_part3[par];
  return s;
}

long n_rows_of(const carel_encoder_args* a) {
  return (a->tok_row && a->n_tokens > 0) ? (long)a->n_tokens : (long)a->batch * a->seq_len;
}

int enc_check(const carel_encoder_args* a, const char* who) {
  if (!a) return set_error(CAREL_ERR_ARG, "%s: null args", who);
  if (a->hidden != EH || a->heads != ENH || a->intermediate != EI)
    return set_error(CAREL_ERR_SHAPE, "%s: only the BERT-base geometry (768/12/3072) is supported", who);
  if (a->batch < 1 || a->n_layers < 1) return set_error(CAREL_ERR_SHAPE, "%s: bad batch / n_layers", who);
  if (a->seq_len < 32 || a->seq_len > 128 || (a->seq_len & 31)) return set_error(CAREL_ERR_SHAPE, "%s: seq_len must be 32/64/96/128", who);
  if (((long)a->batch * a->seq_len) % 128) return set_error(CAREL_ERR_SHAPE, "%s: batch*seq_len must be a multiple of 128 (pad the batch)", who);
  if (!a->input_ids || !a->layers || !a->act) return set_error(CAREL_ERR_ARG, "%s: null tensor", who);
  if (a->n_cls) {
    if (!a->cls_rows || !a->cls_orig_rows || a->n_cls % 128 || a->n_cls < a->batch || a->n_cls > ((a->batch + 127) / 128) * 128)
      return set_error(CAREL_ERR_ARG, "%s: n_cls must be batch rounded up to 128, with cls_rows and cls_orig_rows", who);
  }
  if (a->tok_row || a->cu_seqlens || a->n_tokens) {
    if (!a->tok_row || !a->cu_seqlens || a->n_tokens <= 0 || a->n_tokens % 128 || a->n_tokens > a->batch * a->seq_len)
      return set_error(CAREL_ERR_ARG, "%s: packing needs tok_row, cu_seqlens and n_tokens (multiple of 128, <= batch*seq_len)", who);
  }
  return CAREL_OK;
}

// One low-priority stream + a few events per device, created on first use and kept for the life of the process.
struct SideStream { static constexpr int NEV = 13; hipStream_t stream; hipStream_t aux; hipStream_t peer; hipEvent_t ev[NEV]; bool ok; };
SideStream* side_stream() {
  static SideStream per_dev[16];
  static bool made[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!made[dev]) {
    SideStream& s = per_dev[dev];
    int lo = 0, hi = 0;
    s.ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
#ifdef CAREL_EXPERIMENTS      // A/B of the side streams' priority (experiments build only): CAREL_SIDE_STREAM_PRIORITY = low (default) | normal | high
    if (const char* e = getenv("CAREL_SIDE_STREAM_PRIORITY")) { if (e[0] == 'n') lo = 0; else if (e[0] == 'h') lo = hi; }
#endif
    s.ok = s.ok &&
           hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, lo) == hipSuccess &&     // lo = lowest priority
           hipStreamCreateWithPriority(&s.aux, hipStreamNonBlocking, lo) == hipSuccess &&
           hipStreamCreateWithFlags(&s.peer, hipStreamNonBlocking) == hipSuccess;               // default priority: the second forward chain
    for (int i = 0; i < SideStream::NEV && s.ok; ++i) s.ok = hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming) == hipSuccess;
    made[dev] = true;
  }
  return per_dev[dev].ok ? &per_dev[dev] : nullptr;
}

// lnres != null (residual epilogue): `resid` holds the pre-LayerNorm rows of the LayerNorm whose output is the residual; the epilogue
// recomputes it (carel_gemm_args.resid_ln_*), so that LayerNorm never writes its f32 output
struct LnResid { const void* stats; const void* gamma; const void* beta; };
CAREL_TUNABLE(int, g_ln_slab_fusion, 1);   // (experiments build: hook 270 / 271) slab epilogues of the split-K GEMMs as their own launch / fused into the LayerNorm that follows
int ln_slab_fusion_enabled() { return g_ln_slab_fusion; }
CAREL_TUNABLE(int, g_wgrad_group, 1);       // (experiments build: hook 240 / 241) one GEMM + reduction per weight gradient / the grouped launch
int wgrad_group_enabled() { return g_wgrad_group; }
CAREL_TUNABLE(int, g_ln_resid, 1);          // tuning hook (carel_gemm_set_variant(230 / 231)): LayerNorm f32 outputs written and read back / recomputed by the next epilogue
int ln_resid_enabled() { return g_ln_resid; }
int gemm_call(const void* A, const void* B, long lda, long ldb, int M, int N, int K, int form, int epi, int splits, void* out_bf16,
              void* out2, void* out_f32, const void* bias, const void* resid, const void* aux, uint32_t seed, uint32_t site,
              uint32_t off, float p, void* stream, void* colsum_part = nullptr, const void* row_map = nullptr, void* ws = nullptr,
              size_t ws_bytes = 0, int split_tile_factor = 1, const LnResid* lnres = nullptr, int* plan = nullptr) {
  carel_gemm_args g;
  g.resid_ln_stats = lnres ? lnres->stats : nullptr; g.resid_ln_gamma = lnres ? lnres->gamma : nullptr; g.resid_ln_beta = lnres ? lnres->beta : nullptr;
  g.A = A; g.B = B; g.lda = lda; g.ldb = ldb; g.ldc = N; g.M = M; g.N = N; g.K = K; g.form = form; g.epilogue = epi; g.splits = splits;
  g.out_bf16 = out_bf16; g.out2_bf16 = out2; g.out_f32 = out_f32; g.bias = bias; g.resid_f32 = resid; g.aux_bf16 = aux;
  g.drop_seed = seed; g.drop_site = site; g.drop_idx_offset = off; g.drop_p = p; g.colsum_part = colsum_part; g.drop_row_map = row_map; g.colsum_a = nullptr;
  g.splitk_ws = ws; g.splitk_ws_bytes = (int64_t)ws_bytes;
  // the scratch block is zero-filled when it is allocated (carel_encoder_scratch_bytes: "zero it once") and its workspace's last 4 KiB are
  // written by nobody else: pair split-K may keep its flags there.  Not for the second forward chain, whose workspace is the
  // weight-gradient slab area.
  g.splitk_ws_zeroed = (ws != nullptr && (split_tile_factor & 0xff) == 1) ? 1 : 0;
  if (plan) { *plan = gemm_bf16_split_plan(&g, split_tile_factor); return CAREL_OK; }      // no launch: how many slabs would this call write?
  return gemm_bf16_ex(&g, split_tile_factor, stream);
}

// dW[M,N] = A^T[M x T] * B[T x N]  (A = dY [T,M], B = X [T,N])  via split-K slabs.  With db != null the bias gradient
// db[M] = column sums of dY comes out of the same GEMM (ones-vector MFMA) as [splits][M] partials behind the slabs.
// lnp != null: the partials of the LayerNorm backward that produced dY are summed (-> dgamma, dbeta, dbias) by the SAME launch that reduces
// the slabs (they are independent of the GEMM; one launch instead of three behind every weight gradient).
struct LnPartials { const void* partials; long rows; void* dgamma; void* dbeta; void* dbias; };
int wgrad_call(const void* dY, const void* X, long T, int M, int N, void* slabs, void* dW, void* stream, void* db = nullptr,
               const LnPartials* lnp = nullptr) {
  const int splits = wgrad_splits(T, M, N);
  carel_gemm_args g;
  g.A = dY; g.B = X; g.lda = M; g.ldb = N; g.ldc = N; g.M = M; g.N = N; g.K = (int)T; g.form = CAREL_GEMM_TN; g.epilogue = CAREL_EPI_SLAB_F32;
  g.splits = splits; g.out_bf16 = nullptr; g.out2_bf16 = nullptr; g.out_f32 = slabs; g.bias = nullptr; g.resid_f32 = nullptr; g.aux_bf16 = nullptr;
  g.drop_seed = 0; g.drop_site = 0; g.drop_idx_offset = 0; g.drop_p = 0.f; g.drop_row_map = nullptr; g.colsum_part = nullptr;
  g.splitk_ws = nullptr; g.splitk_ws_bytes = 0; g.splitk_ws_zeroed = 0;
  g.resid_ln_stats = nullptr; g.resid_ln_gamma = nullptr; g.resid_ln_beta = nullptr;
  float* cs = db ? (float*)slabs + (size_t)splits * M * N : nullptr;
  g.colsum_a = cs;
  if (splits == 1) {                 // one slab IS the result: write it (and the bias sums) in place, nothing to reduce
    g.out_f32 = dW;
    if (db) g.colsum_a = db;
    int rc = carel_gemm_bf16(&g, stream);
    if (!rc && lnp) rc = layernorm_bwd_reduce(lnp->partials, lnp->rows, lnp->dgamma, lnp->dbeta, lnp->dbias, (hipStream_t)stream);
    return rc;
  }
  int rc = carel_gemm_bf16(&g, stream);
  if (rc) return rc;
  return slab_reduce_multi(slabs, dW, (int64_t)M * N, cs, db, M, splits, lnp ? lnp->partials : nullptr,
                           lnp ? carel_layernorm_bwd_blocks(lnp->rows) : 0, lnp ? lnp->dgamma : nullptr, lnp ? lnp->dbeta : nullptr,
                           lnp ? lnp->dbias : nullptr, (hipStream_t)stream);
}

}  // namespace

#ifdef CAREL_EXPERIMENTS
namespace carel { void encoder_ln_resid_enable(int on) { g_ln_resid = on ? 1 : 0; } void encoder_wgrad_group_enable(int on) { g_wgrad_group = on ? 1 : 0; }
                  void encoder_ln_slab_fusion_enable(int on) { g_ln_slab_fusion = on ? 1 : 0; } }
#endif

extern "C" void* carel_side_stream(int32_t which) {
  SideStream* sd = side_stream();
  if (!sd) { set_error(CAREL_ERR_HIP, "carel_side_stream: could not create the side streams"); return nullptr; }
  return which == 0 ? (void*)sd->stream : (void*)sd->aux;
}

extern "C" int64_t carel_encoder_act_bytes(int32_t batch, int32_t seq_len, int32_t n_layers, int32_t inference) {
  return (int64_t)act_layout(batch, seq_len, n_layers, inference).total;
}
extern "C" int64_t carel_encoder_scratch_bytes(int32_t batch, int32_t seq_len) {
  return (int64_t)scratch_layout(batch, seq_len).total;
}
extern "C" void* carel_encoder_x_last(const carel_encoder_args* a) {
  if (!a || !a->act) return nullptr;
  const ActLayout l = act_layout(a->batch, a->seq_len, a->n_layers, a->inference);
  return (char*)a->act + l.o_xa;
}

static carel_embed_args embed_args_of(const carel_encoder_args* a, const ActLayout& l, const LayerAct& first) {
  carel_embed_args e;
  e.input_ids = a->input_ids; e.token_type_ids = a->token_type_ids; e.word_emb = a->word_emb; e.pos_emb = a->pos_emb;
  e.type_emb = a->type_emb; e.ln_gamma = a->emb_ln_g; e.ln_beta = a->emb_ln_b; e.ln_eps = a->ln_eps;
  e.batch = a->batch; e.seq_len = a->seq_len; e.hidden = EH; e.vocab_size = a->vocab_size; e.max_pos = a->max_pos;
  e.type_vocab = a->type_vocab; e.roberta = a->roberta; e.pad_id = a->pad_id;
  e.drop_seed = a->drop_seed; e.drop_idx_offset = a->drop_row_offset * (uint32_t)(a->seq_len * EH); e.drop_p = a->hidden_dropout;
  e.x_f32 = (char*)a->act + l.o_xa; e.x_bf16 = first.xin_bf16; e.stats = (char*)a->act + l.o_embst;
  e.tok_row = a->tok_row; e.n_rows = a->tok_row ? a->n_tokens : 0;
  return e;
}

// Layers [l0, l1) of the forward pass for the samples [b0, b0 + nb).  Dense mode: their rows [b0*S, (b0+nb)*S) of every
// activation buffer; packed mode and the [CLS]-only last layer exist for the whole batch only (b0 = 0, nb = batch).
static int forward_layers(const carel_encoder_args* a, int l0, int l1, long b0, long nb, void* stream, char* ws, size_t ws_bytes, int chains = 1) {
  int rc;
  const long B = a->batch, S = a->seq_len;
  const bool whole = b0 == 0 && nb == B;
  const long T = whole ? n_rows_of(a) : nb * S;                  // rows processed by this call
  const long r0 = b0 * S;                                        // first row (dense)
  const ActLayout l = act_layout(B, S, a->n_layers, a->inference);
  char* base = (char*)a->act;
  char* xa = base + l.o_xa + (size_t)r0 * EH * 4;
  char* xb = base + l.o_xb + (size_t)r0 * EH * 4;
  const uint32_t hoff = (a->drop_row_offset + (uint32_t)b0) * (uint32_t)(S * EH), aoff = (a->drop_row_offset + (uint32_t)b0) * (uint32_t)(ENH * S * S);
  for (int i = l0; i < l1; ++i) {
    const carel_layer_params& w = a->layers[i];
    LayerAct la = layer_act(l, base, i, a->inference);
    la.xin_bf16 += (size_t)r0 * EH * 2; la.qkv += (size_t)r0 * 3 * EH * 2; la.lse += (size_t)b0 * ENH * S * 4; la.ctx += (size_t)r0 * EH * 2;
    la.h1 += (size_t)r0 * EH * 4; la.st1 += (size_t)r0 * 2 * 4; la.x1_bf16 += (size_t)r0 * EH * 2; la.u += (size_t)r0 * EI * 2;
    la.g += (size_t)r0 * EI * 2; la.h2 += (size_t)r0 * EH * 4; la.st2 += (size_t)r0 * 2 * 4;
    if ((rc = gemm_call(la.xin_bf16, w.qkv_w, EH, EH, (int)T, 3 * EH, EH, CAREL_GEMM_NT, CAREL_EPI_BIAS_BF16, 1, la.qkv, nullptr, nullptr,
                        w.qkv_b, nullptr, nullptr, 0, 0, 0, 0.f, stream, nullptr, nullptr, ws, ws_bytes, chains))) return rc;
    carel_attn_args at;
    at.qkv = la.qkv; at.attention_mask = a->attention_mask ? (const void*)((const long*)a->attention_mask + b0 * S) : nullptr;
    at.ctx = la.ctx; at.lse = la.lse; at.dctx = nullptr; at.dqkv = nullptr;
    at.batch = (int)nb; at.seq_len = (int)S; at.heads = ENH; at.head_dim = 64;
    at.drop_seed = a->drop_seed; at.drop_site = 1 + 3 * i; at.drop_idx_offset = aoff; at.drop_p = a->attn_dropout;
    at.cu_seqlens = whole ? a->cu_seqlens : nullptr;
    at.rel_bias_dist = a->rel_bias_dist; at.d_rel_bias_dist = nullptr;
    const bool cls_only = a->n_cls > 0 && i + 1 == a->n_layers;
    // the [CLS]-only last layer reads one context row per sample (position 0): the other query tiles are never computed
    at.q_rows = cls_only ? 32 : 0;
    if ((rc = carel_attention_fwd(&at, stream))) return rc;
    if (cls_only && !whole) return set_error(CAREL_ERR_ARG, "carel_encoder_forward: internal: [CLS]-only layer on a partial batch");
    long R = T;                                  // rows of the row-wise half of this layer
    const int fixed = cls_only ? GEMM_EX_FIXED_ROWS : 0;
    const void* Actx = la.ctx; const void* res1 = xa; const void* rmap = whole ? a->tok_row : nullptr;
    // The f32 output of a LayerNorm is only ever read as the residual of the next linear's epilogue: that epilogue recomputes it from the
    // pre-LayerNorm rows and the statistics the backward pass keeps anyway (bit-identical, tests/test_gpu_gemm.py), and the LayerNorm does not
    // write it (25 MB per sub-layer at T = 8192).  Exceptions, which still get their f32 rows: the embedding output (layer 0's residual),
    // the input of a [CLS]-only last layer (gathered by row) and the encoder's final output.  g_ln_resid: tuning hook 230 / 231.
    LnResid lr1, lr2;
    const LnResid* plr1 = nullptr;               // residual of the out-projection = LayerNorm 2 of the previous layer
    const bool lnres_on = ln_resid_enabled() && !gemm_rowln_wanted(T);      // (the fused row-band kernel writes its own f32 rows)
    if (lnres_on && i > 0 && !cls_only) {
      LayerAct lp = layer_act(l, base, i - 1, a->inference);
      lr1.stats = (const char*)lp.st2 + (size_t)r0 * 2 * 4; lr1.gamma = a->layers[i - 1].ln2_g; lr1.beta = a->layers[i - 1].ln2_b;
      res1 = (const char*)lp.h2 + (size_t)r0 * EH * 4; plr1 = &lr1;
    }
    if (cls_only) {                              // only the [CLS] rows of the last layer are ever read
      R = a->n_cls;
      char* cctx = base + l.o_cctx; char* cxres = base + l.o_cxres;
      if ((rc = gather_rows(xa, la.ctx, a->cls_rows, (int)R, cxres, cctx, (hipStream_t)stream))) return rc;
      Actx = cctx; res1 = cxres; rmap = a->cls_orig_rows;
    }
    const bool lnres2 = lnres_on;                // residual of FFN2 = LayerNorm 1 of this layer
    lr2.stats = la.st1; lr2.gamma = w.ln1_g; lr2.beta = w.ln1_b;
    // does anything read this layer's f32 output rows?  (the next layer's epilogue recomputes them unless it is a [CLS]-only last layer)
    const bool need_xa = !lnres_on || i + 1 == a->n_layers || (a->n_cls > 0 && i + 2 == a->n_layers);
    // dense batches: linear + dropout + residual + LayerNorm as ONE kernel (gemm_rowln.hip: 32 complete rows per workgroup; the same bits
    // as the GEMM followed by the stand-alone LayerNorm).  Packed ECPE batches and the [CLS]-only rows are too few rows to fill the chip
    // with 32-row workgroups that each stream the whole weight matrix: they keep the two-kernel path.
    const bool fuse_ln = gemm_rowln_wanted(R);
    auto linear_ln = [&](const void* A, const void* W, int K, const void* bias, const void* resid, int site, const void* g, const void* bt,
                         void* h, void* xf, void* xbf, void* st) -> int {
#ifndef CAREL_EXPERIMENTS
      return set_error(CAREL_ERR_ARG, "carel_encoder_forward: internal: the fused row-band kernel exists in the experiments build only");
#else
      carel_gemm_rowln_args ra;
      ra.A = A; ra.W = W; ra.lda = K; ra.ldb = K; ra.M = (int)R; ra.K = K; ra.bias = bias; ra.resid_f32 = resid; ra.gamma = g; ra.beta = bt; ra.eps = a->ln_eps;
      ra.h_f32 = h; ra.x_f32 = xf; ra.x_bf16 = xbf; ra.stats = st;
      ra.w_packed = 0;
      ra.drop_seed = a->drop_seed; ra.drop_site = (uint32_t)site; ra.drop_idx_offset = hoff; ra.drop_p = a->hidden_dropout; ra.drop_row_map = rmap;
      return carel_gemm_rowln(&ra, stream);
#endif
    };
    if (fuse_ln) {
      if ((rc = linear_ln(Actx, w.out_w, EH, w.out_b, res1, 2 + 3 * i, w.ln1_g, w.ln1_b, a->inference ? nullptr : la.h1, xb, la.x1_bf16, la.st1))) return rc;
    } else {
    // (packed batches: where the linear runs split-K into slabs, its slab epilogue -- bias, dropout, residual -- is the first half of the LayerNorm
    // kernel behind it: one launch and one round trip of the rows fewer per sub-layer; ln_fwd_slabs_kernel, same expressions in the same order)
    int out_slabs = 1;
    if (ln_slab_fusion_enabled() && ws &&
        (rc = gemm_call(Actx, w.out_w, EH, EH, (int)R, EH, EH, CAREL_GEMM_NT, CAREL_EPI_BIAS_DROP_RESID, 1, nullptr, nullptr, la.h1, w.out_b, res1, nullptr,
                        a->drop_seed, 2 + 3 * i, hoff, a->hidden_dropout, stream, nullptr, rmap, ws, ws_bytes, chains | fixed, plr1, &out_slabs))) return rc;
    if ((rc = gemm_call(Actx, w.out_w, EH, EH, (int)R, EH, EH, CAREL_GEMM_NT, CAREL_EPI_BIAS_DROP_RESID, 1, nullptr, nullptr, la.h1,
                        w.out_b, res1, nullptr, a->drop_seed, 2 + 3 * i, hoff, a->hidden_dropout, stream, nullptr, rmap, ws, ws_bytes,
                        chains | fixed | (out_slabs > 1 ? GEMM_EX_DEFER_EPILOGUE : 0), plr1))) return rc;
    if (out_slabs > 1) {
      if ((rc = layernorm_fwd_slabs(ws, out_slabs, w.out_b, res1, plr1 ? plr1->stats : nullptr, plr1 ? plr1->gamma : nullptr, plr1 ? plr1->beta : nullptr,
                                    a->drop_seed, 2 + 3 * i, hoff, a->hidden_dropout, rmap, la.h1, w.ln1_g, w.ln1_b, a->ln_eps, R, lnres2 ? nullptr : xb,
                                    la.x1_bf16, la.st1, (hipStream_t)stream))) return rc;
    } else if ((rc = carel_layernorm_fwd(la.h1, w.ln1_g, w.ln1_b, a->ln_eps, R, EH, lnres2 ? nullptr : xb, la.x1_bf16, la.st1, stream))) return rc;
    }
    if ((rc = gemm_call(la.x1_bf16, w.ffn1_w, EH, EH, (int)R, EI, EH, CAREL_GEMM_NT, a->inference ? CAREL_EPI_BIAS_GELU : CAREL_EPI_BIAS_GELU_DG, 1, a->inference ? nullptr : la.u, la.g, nullptr,
                        w.ffn1_b, nullptr, nullptr, 0, 0, 0, 0.f, stream, nullptr, nullptr, ws, ws_bytes, chains | fixed))) return rc;
    char* next_bf16 = nullptr;
    if (i + 1 < a->n_layers) next_bf16 = layer_act(l, base, i + 1, a->inference).xin_bf16 + (size_t)r0 * EH * 2;
    if (fuse_ln && gemm_rowln_wanted_k(EI)) {
      if ((rc = linear_ln(la.g, w.ffn2_w, EI, w.ffn2_b, xb, 3 + 3 * i, w.ln2_g, w.ln2_b, a->inference ? nullptr : la.h2, xa, next_bf16, la.st2))) return rc;
    } else {
    const void* res2 = lnres2 ? (const void*)la.h1 : (const void*)xb;
    int ffn2_slabs = 1;
    if (ln_slab_fusion_enabled() && ws &&
        (rc = gemm_call(la.g, w.ffn2_w, EI, EI, (int)R, EH, EI, CAREL_GEMM_NT, CAREL_EPI_BIAS_DROP_RESID, 1, nullptr, nullptr, la.h2, w.ffn2_b, res2, nullptr,
                        a->drop_seed, 3 + 3 * i, hoff, a->hidden_dropout, stream, nullptr, rmap, ws, ws_bytes, chains | fixed, lnres2 ? &lr2 : nullptr, &ffn2_slabs))) return rc;
    if ((rc = gemm_call(la.g, w.ffn2_w, EI, EI, (int)R, EH, EI, CAREL_GEMM_NT, CAREL_EPI_BIAS_DROP_RESID, 1, nullptr, nullptr, la.h2,
                        w.ffn2_b, res2, nullptr, a->drop_seed, 3 + 3 * i, hoff, a->hidden_dropout, stream, nullptr, rmap, ws,
                        ws_bytes, chains | fixed | (ffn2_slabs > 1 ? GEMM_EX_DEFER_EPILOGUE : 0), lnres2 ? &lr2 : nullptr))) return rc;
    if (ffn2_slabs > 1) {
      if ((rc = layernorm_fwd_slabs(ws, ffn2_slabs, w.ffn2_b, res2, lnres2 ? lr2.stats : nullptr, lnres2 ? lr2.gamma : nullptr, lnres2 ? lr2.beta : nullptr,
                                    a->drop_seed, 3 + 3 * i, hoff, a->hidden_dropout, rmap, la.h2, w.ln2_g, w.ln2_b, a->ln_eps, R, need_xa ? xa : nullptr,
                                    next_bf16, la.st2, (hipStream_t)stream))) return rc;
    } else if ((rc = carel_layernorm_fwd(la.h2, w.ln2_g, w.ln2_b, a->ln_eps, R, EH, need_xa ? xa : nullptr, next_bf16, la.st2, stream))) return rc;
    }
  }
  return CAREL_OK;
}

extern "C" int carel_encoder_forward(const carel_encoder_args* a, void* stream) {
  int rc = enc_check(a, "carel_encoder_forward");
  if (rc) return rc;
  if (!a->word_emb || !a->pos_emb || !a->type_emb || !a->emb_ln_g || !a->emb_ln_b) return set_error(CAREL_ERR_ARG, "carel_encoder_forward: null embedding tensor");
  const long B = a->batch, S = a->seq_len;
  const ActLayout l = act_layout(B, S, a->n_layers, a->inference);
  const ScratchLayout sl = scratch_layout(B, S);
  char* ws = a->scratch ? (char*)a->scratch + sl.o_ws : nullptr;          // split-K workspace for small (packed) batches
  const size_t ws_bytes = a->scratch ? sl.ws_bytes : 0;
  LayerAct la = layer_act(l, (char*)a->act, 0, a->inference);
  carel_embed_args e = embed_args_of(a, l, la);
  if (!a->inference && a->scratch && embed_sort_supported(&e)) {
    // a backward pass will follow: leave every row's (token id, position id) behind and sort them now -- on the weight-gradient side
    // stream when there is one (idle during the forward pass; carel_encoder_backward_embeddings joins it), so that the embedding tables'
    // gradients come from fixed-order segment sums instead of float atomics without a sort on the backward pass's critical path
    void* sort_ws = (char*)a->scratch + sl.o_sort;
    if ((rc = embed_ln_fwd_keys(&e, sort_ws, (hipStream_t)stream))) return rc;
    // (not beside the two forward chains: a third stream at work during the forward pass made that mode 1.8x slower -- HIP maps streams
    // onto four hardware queues)
    SideStream* ss = ((a->overlap_wgrad & 1) && !(a->overlap_wgrad & 2)) ? side_stream() : nullptr;
    if (ss) {
      if (hipEventRecord(ss->ev[8], (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(ss->stream, ss->ev[8], 0) != hipSuccess)
        return set_error(CAREL_ERR_HIP, "carel_encoder_forward: event fork failed");
      if ((rc = embed_sort_rows(&e, sort_ws, ss->stream))) return rc;
    } else if ((rc = embed_sort_rows(&e, sort_ws, (hipStream_t)stream))) return rc;
  } else if ((rc = carel_embed_ln_fwd(&e, stream))) return rc;
  // Samples are independent through the whole encoder: with a->overlap_wgrad (dense rows, even batch) the two halves of
  // the batch run as two chains, one on `stream`, one on a peer stream of the same priority, so that each chain's launch gaps, tile-count
  // tails and memory-bound kernels are filled by the other's GEMMs.  The (optionally [CLS]-only) last layer runs whole.
  const long hb = B / 2;
  const int lsplit = a->n_cls > 0 ? a->n_layers - 1 : a->n_layers;
  SideStream* sd = nullptr;
  if ((a->overlap_wgrad & 2) && !a->tok_row && (B & 1) == 0 && (hb * S) % 128 == 0 && lsplit >= 1 && a->scratch) sd = side_stream();
  if (!sd) return forward_layers(a, 0, a->n_layers, 0, B, stream, ws, ws_bytes);
  if (hipEventRecord(sd->ev[0], (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(sd->peer, sd->ev[0], 0) != hipSuccess)
    return set_error(CAREL_ERR_HIP, "carel_encoder_forward: event fork failed");
  char* ws2 = (char*)a->scratch + sl.o_slabs;                    // the weight-gradient slabs are idle during the forward pass
  // chains = 2: same split-K choices (same bits) as the one-chain forward
  rc = forward_layers(a, 0, lsplit, hb, hb, (void*)sd->peer, ws2, sl.ws_bytes, 2);
  if (!rc) rc = forward_layers(a, 0, lsplit, 0, hb, stream, ws, ws_bytes, 2);
  if (rc) return rc;
  if (hipEventRecord(sd->ev[1], sd->peer) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, sd->ev[1], 0) != hipSuccess)
    return set_error(CAREL_ERR_HIP, "carel_encoder_forward: event join failed");
  if (lsplit < a->n_layers) return forward_layers(a, lsplit, a->n_layers, 0, B, stream, ws, ws_bytes);
  return CAREL_OK;
}

// Backward of encoder layer `layer`.  a->dx holds d(loss)/d(layer output) on entry and d(loss)/d(layer
// input) on return.  Writes every gradient of a->layer_grads[layer].
extern "C" int carel_encoder_backward_layer(const carel_encoder_args* a, int32_t layer, void* stream) {
  int rc = enc_check(a, "carel_encoder_backward_layer");
  if (rc) return rc;
  if (a->inference) return set_error(CAREL_ERR_ARG, "carel_encoder_backward_layer: forward ran in inference mode (no saved activations)");
  if (layer < 0 || layer >= a->n_layers || !a->layer_grads || !a->scratch || !a->dx)
    return set_error(CAREL_ERR_ARG, "carel_encoder_backward_layer: bad layer index or null buffer");
  const long B = a->batch, S = a->seq_len, T = n_rows_of(a);
  const ActLayout l = act_layout(B, S, a->n_layers, 0);
  const LayerAct la = layer_act(l, (char*)a->act, layer, 0);
  const ScratchLayout sl = scratch_layout(B, S);
  const int par = layer & 1;                   // the weight-gradient operands live in the buffer set of the layer's parity (scratch_layout)
  const Scratch s = scratch_of(sl, (char*)a->scratch, par);
  const carel_layer_params& w = a->layers[layer];
  const carel_layer_grads& g = a->layer_grads[layer];
  const uint32_t hoff = a->drop_row_offset * (uint32_t)(S * EH), aoff = a->drop_row_offset * (uint32_t)(ENH * S * S);
  const size_t ws_bytes = sl.ws_bytes;
  // Weight gradients on a second stream (a->overlap_wgrad): each dW GEMM + slab reduction is forked (events ev[0..3])
  // behind the kernel that produced its dY operand and runs beside the data-gradient chain, LayerNorm and attention
  // backward kernels.  There is no join at the end of the layer: the side stream records ev[4..7] after each of its four
  // groups, and the main stream waits for group i of the PREVIOUS call only just before it overwrites the scratch
  // buffers that group read (dyb/part, du/part2, dyb2/part3, dqkv).  Consequently, in stream order after this call
  // returns, the gradients of layer+1 are complete (the main stream waited for its last group before this layer's
  // attention backward), those of `layer` after the next call or carel_encoder_backward_join.
  SideStream* sd = nullptr;
  if (a->overlap_wgrad & 1) {
    sd = side_stream();
    if (!sd) return set_error(CAREL_ERR_HIP, "carel_encoder_backward_layer: could not create the side stream / events");
  }
  void* wstream = sd ? (void*)sd->stream : stream;
  int nfork = 0;
  auto fork = [&]() -> int {      // side stream waits for everything enqueued on `stream` so far
    if (!sd) return CAREL_OK;
    hipEvent_t ev = sd->ev[nfork++];
    if (hipEventRecord(ev, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(sd->stream, ev, 0) != hipSuccess)
      return set_error(CAREL_ERR_HIP, "carel_encoder_backward_layer: event fork failed");
    return CAREL_OK;
  };
  auto group_done = [&](int i) -> int {       // side stream: group i of this layer is enqueued
    if (!sd) return CAREL_OK;
    return hipEventRecord(sd->ev[4 + i], sd->stream) == hipSuccess ? CAREL_OK : set_error(CAREL_ERR_HIP, "carel_encoder_backward_layer: event record failed");
  };
  auto wait_group = [&](int i) -> int {       // main stream: group i of the previous call has finished with its buffers
    if (!sd) return CAREL_OK;                 // (an event that was never recorded counts as complete)
    return hipStreamWaitEvent((hipStream_t)stream, sd->ev[4 + i], 0) == hipSuccess ? CAREL_OK : set_error(CAREL_ERR_HIP, "carel_encoder_backward_layer: event wait failed");
  };
  // Row-wise half of the layer.  Last layer with dead-row elimination: only the n_cls [CLS] rows exist (compact).
  const bool cls_only = a->n_cls > 0 && layer + 1 == a->n_layers;
  const long R = cls_only ? (long)a->n_cls : T;
  const void* rmap = cls_only ? a->cls_orig_rows : a->tok_row;
  const void* ctx_rows = cls_only ? (const void*)((char*)a->act + l.o_cctx) : (const void*)la.ctx;
  // The layer's four weight gradients (+ the two bias gradients that ride on them, + the sums of both LayerNorm backward passes' partials)
  // as ONE grouped launch behind the attention backward: carel_gemm_wgrad_group (whole output tiles written in place, only the
  // remainder of the tile count goes through split-K partials: 25 MB per layer instead of 165 MB of slabs, 2 launches instead of 8).
  // Not for a [CLS]-only last layer (its row-wise half has 128 rows: two K tiles) or token counts below the kernel's minimum: those keep
  // one split-K GEMM + reduction per weight gradient, forked behind the kernel that produced its dY.
  carel_wgrad_group_args ga;
  ga.prob[0] = carel_wgrad_problem{s.dyb, la.g, g.ffn2_w, nullptr, EH, EI};
  ga.prob[1] = carel_wgrad_problem{s.du, la.x1_bf16, g.ffn1_w, g.ffn1_b, EI, EH};
  ga.prob[2] = carel_wgrad_problem{s.dqkv, la.xin_bf16, g.qkv_w, g.qkv_b, 3 * EH, EH};
  ga.prob[3] = carel_wgrad_problem{s.dyb2, la.ctx, g.out_w, nullptr, EH, EH};
  ga.n_prob = 4; ga.T = T; ga.workspace = s.slabs; ga.workspace_bytes = (int64_t)sl.slab_bytes;
  ga.ln[0] = carel_ln_partial_set{s.part, R, g.ln2_g, g.ln2_b, g.ffn2_b};
  ga.ln[1] = carel_ln_partial_set{s.part3, R, g.ln1_g, g.ln1_b, g.out_b};
  ga.n_ln = 2;
  bool grouped = false;
  if (wgrad_group_enabled() && !cls_only) {
    const int64_t need = carel_gemm_wgrad_group_ws_bytes(&ga);
    grouped = need >= 0 && need <= (int64_t)sl.slab_bytes;
  }
  // Packed ECPE batches (~1.8 k rows): the N = 768 data-gradient GEMMs run split-K into slabs + a slab epilogue.  Where the only reader of that
  // epilogue's f32 output is the LayerNorm backward that follows (FFN1 data gradient -> LayerNorm 1; QKV data gradient -> LayerNorm 2 of the
  // layer below, i.e. the NEXT call), the epilogue is deferred into that kernel (GEMM_EX_DEFER_EPILOGUE, layernorm_bwd_rows_slabs): same bits,
  // one launch and one round trip of the rows fewer per sub-layer.  The decision is a function of the shapes only, so this call knows what
  // the previous one did.  Layer 0's QKV data gradient feeds the embedding backward and is never deferred.
  int qkv_slabs = 1;
  if (ln_slab_fusion_enabled()) {
    if ((rc = gemm_call(s.dqkv, w.qkv_w, 3 * EH, EH, (int)T, EH, 3 * EH, CAREL_GEMM_NN, CAREL_EPI_ADD_F32, 1, nullptr, nullptr, a->dx, nullptr,
                        s.dy, nullptr, 0, 0, 0, 0.f, stream, nullptr, nullptr, s.ws, ws_bytes, 1, nullptr, &qkv_slabs))) return rc;
  }
  // (did the call for layer + 1 defer its QKV epilogue?  then dx = sum of the slabs + s.dy; a [CLS]-only last layer never defers)
  const bool dx_in_slabs = qkv_slabs > 1 && layer + 1 < a->n_layers && !(a->n_cls > 0 && layer + 2 == a->n_layers);
  // LN2 backward: dx -> dh2 (s.dy), dyb (dropout-masked, bf16), dgamma/dbeta, FFN2 bias grad
  if ((rc = wait_group(0))) return rc;
  if (dx_in_slabs) {
    if ((rc = layernorm_bwd_rows_slabs(s.ws, qkv_slabs, s.dy, la.h2, la.st2, w.ln2_g, R, a->drop_seed, 3 + 3 * layer, hoff, a->hidden_dropout, rmap, s.dy,
                                       s.dyb, s.part, (hipStream_t)stream))) return rc;
  } else if ((rc = layernorm_bwd_rows(a->dx, la.h2, la.st2, w.ln2_g, R, a->drop_seed, 3 + 3 * layer, hoff, a->hidden_dropout, rmap, s.dy, s.dyb,
                               s.part, (hipStream_t)stream))) return rc;
  // FFN2: du = (dyb W2) * gelu'(u) ; dW2 = dyb^T g
  //       the FFN1 bias gradient (column sums of du) comes out of the same epilogue as per-row-tile partials
  if (!grouped) {
    if ((rc = fork())) return rc;
    const LnPartials lp2{s.part, R, g.ln2_g, g.ln2_b, g.ffn2_b};
    if ((rc = wgrad_call(s.dyb, la.g, R, EH, EI, s.slabs, g.ffn2_w, wstream, nullptr, &lp2))) return rc;
    if ((rc = group_done(0))) return rc;
  }
  if ((rc = wait_group(1))) return rc;
  if ((rc = gemm_call(s.dyb, w.ffn2_w, EH, EI, (int)R, EI, EH, CAREL_GEMM_NN, CAREL_EPI_MUL_BF16, 1, s.du, nullptr, nullptr, nullptr,
                      nullptr, la.u, 0, 0, 0, 0.f, stream, nullptr))) return rc;
  // FFN1: dx1 = du W1 + dh2 -> a->dx ; dW1 = du^T x1, and the FFN1 bias gradient (column sums of du) from the same GEMM: its ones-vector
  // MFMAs are free there (43.2 vs 43.1 us, tools/bench_wgrad_colsum.py), the fused column sums of the data-gradient epilogue cost 3.9 us
  if (!grouped) {
    if ((rc = fork())) return rc;
    if ((rc = wgrad_call(s.du, la.x1_bf16, R, EI, EH, s.slabs, g.ffn1_w, wstream, g.ffn1_b))) return rc;
    if ((rc = group_done(1))) return rc;
  }
  int ffn1_slabs = 1;
  if (ln_slab_fusion_enabled() && !cls_only) {
    if ((rc = gemm_call(s.du, w.ffn1_w, EI, EH, (int)R, EH, EI, CAREL_GEMM_NN, CAREL_EPI_ADD_F32, 1, nullptr, nullptr, a->dx, nullptr, s.dy,
                        nullptr, 0, 0, 0, 0.f, stream, nullptr, nullptr, s.ws, ws_bytes, 1, nullptr, &ffn1_slabs))) return rc;
  }
  if ((rc = gemm_call(s.du, w.ffn1_w, EI, EH, (int)R, EH, EI, CAREL_GEMM_NN, CAREL_EPI_ADD_F32, 1, nullptr, nullptr, a->dx, nullptr, s.dy,
                      nullptr, 0, 0, 0, 0.f, stream, nullptr, nullptr, s.ws, ws_bytes,
                      1 | (cls_only ? GEMM_EX_FIXED_ROWS : 0) | (ffn1_slabs > 1 ? GEMM_EX_DEFER_EPILOGUE : 0)))) return rc;
  // LN1 backward (its bf16 output goes to a second buffer: the FFN2 weight gradient may still be reading dyb)
  if ((rc = wait_group(2))) return rc;
  if (ffn1_slabs > 1) {          // dx = sum of the FFN1 data gradient's slabs + dh2 (s.dy), never stored
    if ((rc = layernorm_bwd_rows_slabs(s.ws, ffn1_slabs, s.dy, la.h1, la.st1, w.ln1_g, R, a->drop_seed, 2 + 3 * layer, hoff, a->hidden_dropout, rmap, s.dy,
                                       s.dyb2, s.part3, (hipStream_t)stream))) return rc;
  } else if ((rc = layernorm_bwd_rows(a->dx, la.h1, la.st1, w.ln1_g, R, a->drop_seed, 2 + 3 * layer, hoff, a->hidden_dropout, rmap, s.dy, s.dyb2,
                               s.part3, (hipStream_t)stream))) return rc;
  // out-proj: dctx = dyb Wo ; dWo = dyb^T ctx
  void* dctx_rows = cls_only ? (void*)s.dqkv : (void*)s.dctx;     // compact result parks in the (still free) dqkv buffer
  if (!grouped) {
    if ((rc = fork())) return rc;
    const LnPartials lp1{s.part3, R, g.ln1_g, g.ln1_b, g.out_b};
    if ((rc = wgrad_call(s.dyb2, ctx_rows, R, EH, EH, s.slabs, g.out_w, wstream, nullptr, &lp1))) return rc;
    if ((rc = group_done(2))) return rc;
  }
  if (cls_only && (rc = wait_group(3))) return rc;             // the compact result parks in dqkv
  if ((rc = gemm_call(s.dyb2, w.out_w, EH, EH, (int)R, EH, EH, CAREL_GEMM_NN, CAREL_EPI_BIAS_BF16, 1, dctx_rows, nullptr, nullptr, nullptr,
                      nullptr, nullptr, 0, 0, 0, 0.f, stream, nullptr, nullptr, s.ws, ws_bytes, 1 | (cls_only ? GEMM_EX_FIXED_ROWS : 0)))) return rc;
  if (cls_only) {
    // expand the compact [CLS] gradients to token rows: dctx (attention backward input) and the residual-path
    // gradient dh1 (added by the QKV dgrad epilogue) are zero everywhere else
    hipError_t he = hipMemsetAsync(s.dctx, 0, (size_t)T * EH * 2, (hipStream_t)stream);
    if (he == hipSuccess) he = hipMemsetAsync(a->dx, 0, (size_t)T * EH * 4, (hipStream_t)stream);
    if (he != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_encoder_backward_layer: memset: %s", hipGetErrorString(he));
    if ((rc = scatter_rows(nullptr, s.dqkv, a->cls_rows, (int)R, nullptr, s.dctx, (hipStream_t)stream))) return rc;
    if ((rc = scatter_rows(s.dy, nullptr, a->cls_rows, (int)R, a->dx, nullptr, (hipStream_t)stream))) return rc;
  }
  const void* dh1_full = cls_only ? (const void*)a->dx : (const void*)s.dy;      // in-place residual add is safe (same thread)
  // attention backward
  carel_attn_args at;
  at.qkv = la.qkv; at.attention_mask = a->attention_mask; at.ctx = la.ctx; at.lse = la.lse; at.dctx = s.dctx; at.dqkv = s.dqkv;
  at.batch = (int)B; at.seq_len = (int)S; at.heads = ENH; at.head_dim = 64;
  at.drop_seed = a->drop_seed; at.drop_site = 1 + 3 * layer; at.drop_idx_offset = aoff; at.drop_p = a->attn_dropout;
  at.cu_seqlens = a->cu_seqlens;
  at.rel_bias_dist = a->rel_bias_dist; at.d_rel_bias_dist = a->d_rel_bias_dist;
  at.q_rows = cls_only ? 32 : 0;                 // dctx is zero off the [CLS] rows: only the first query tile carries a gradient
  if (a->tok_row && layer + 2 >= a->n_layers) {
    // packed: the attention backward writes only rows that belong to a sample; the filler rows up to the next multiple
    // of 128 must be exact zeros for the column sums / dgrad / wgrad GEMMs that read dqkv over all T rows.  Once per backward pass and
    // buffer set (the pass's first two calls are the last two layers): nothing else writes those rows between the layers of one pass
    hipError_t he = hipMemsetAsync(s.dqkv + (size_t)(T - 128) * 3 * EH * 2, 0, (size_t)128 * 3 * EH * 2, (hipStream_t)stream);
    if (he != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_encoder_backward_layer: memset: %s", hipGetErrorString(he));
  }
  if ((rc = wait_group(3))) return rc;
  if ((rc = carel_attention_bwd(&at, stream))) return rc;
  // QKV: dx_in = dqkv Wqkv + dh1 -> a->dx ; dWqkv = dqkv^T x_in
  if ((rc = fork())) return rc;
  if (grouped) { if ((rc = carel_gemm_wgrad_group(&ga, wstream))) return rc; }
  else if ((rc = wgrad_call(s.dqkv, la.xin_bf16, T, 3 * EH, EH, s.slabs, g.qkv_w, wstream, g.qkv_b))) return rc;
  // (deferred: the next call's LayerNorm-2 backward adds the slabs and dh1 itself -- dh1 must then be s.dy, i.e. not the [CLS]-only layer's scattered rows)
  const bool defer_qkv = qkv_slabs > 1 && layer > 0 && !cls_only;
  if ((rc = gemm_call(s.dqkv, w.qkv_w, 3 * EH, EH, (int)T, EH, 3 * EH, CAREL_GEMM_NN, CAREL_EPI_ADD_F32, 1, nullptr, nullptr, a->dx, nullptr,
                      dh1_full, nullptr, 0, 0, 0, 0.f, stream, nullptr, nullptr, s.ws, ws_bytes, 1 | (defer_qkv ? GEMM_EX_DEFER_EPILOGUE : 0)))) return rc;
  if (!grouped && (rc = group_done(3))) return rc;
  if (sd) {
    // side stream: this layer's weight gradients are enqueued; main stream: those of the PREVIOUS call (layer + 1) are complete from here
    // on in stream order -- they were enqueued a whole layer ago -- which also frees that call's buffer set for the next call
    if (hipEventRecord(sd->ev[10 + par], sd->stream) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, sd->ev[10 + (par ^ 1)], 0) != hipSuccess)
      return set_error(CAREL_ERR_HIP, "carel_encoder_backward_layer: event record / wait failed");
  }
  return CAREL_OK;
}

// `stream` waits for every weight-gradient kernel enqueued so far (no-op without overlap_wgrad).  Call it after the last
// carel_encoder_backward_layer before using layer 0's gradients; carel_encoder_backward_embeddings calls it itself.
extern "C" int carel_encoder_backward_join(const carel_encoder_args* a, void* stream) {
  if (!a) return set_error(CAREL_ERR_ARG, "carel_encoder_backward_join: null args");
  if (!(a->overlap_wgrad & 1)) return CAREL_OK;
  SideStream* sd = side_stream();
  if (!sd) return set_error(CAREL_ERR_HIP, "carel_encoder_backward_join: no side stream");
  hipEvent_t ev = sd->ev[SideStream::NEV - 1];
  if (hipEventRecord(ev, sd->stream) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, ev, 0) != hipSuccess)
    return set_error(CAREL_ERR_HIP, "carel_encoder_backward_join: event join failed");
  return CAREL_OK;
}

// Backward of the embedding block: a->dx = d(loss)/d(embedding output).  d_word_emb / d_pos_emb are
// zeroed here and then accumulated with float atomics.
extern "C" int carel_encoder_backward_embeddings(const carel_encoder_args* a, void* stream) {
  int rc = enc_check(a, "carel_encoder_backward_embeddings");
  if (rc) return rc;
  if ((rc = carel_encoder_backward_join(a, stream))) return rc;
  if (a->inference || !a->scratch || !a->dx || !a->d_word_emb || !a->d_pos_emb || !a->d_type_emb || !a->d_emb_ln_g || !a->d_emb_ln_b)
    return set_error(CAREL_ERR_ARG, "carel_encoder_backward_embeddings: null buffer or inference-mode forward");
  const long B = a->batch, S = a->seq_len;
  const ActLayout l = act_layout(B, S, a->n_layers, 0);
  const LayerAct la = layer_act(l, (char*)a->act, 0, 0);
  const Scratch s = scratch_of(scratch_layout(B, S), (char*)a->scratch);
  carel_embed_args e = embed_args_of(a, l, la);
  hipError_t he = hipMemsetAsync(a->d_word_emb, 0, (size_t)a->vocab_size * EH * 4, (hipStream_t)stream);
  if (he == hipSuccess) he = hipMemsetAsync(a->d_pos_emb, 0, (size_t)a->max_pos * EH * 4, (hipStream_t)stream);
  if (he != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_encoder_backward_embeddings: memset: %s", hipGetErrorString(he));
  // s.dy (f32 [T, 768]) is free here: scratch for the fixed-order position-table reduction
  // (the sorted keys were made by carel_encoder_forward for this batch: same condition here as there)
  const ScratchLayout sl = scratch_layout(B, S);
  const void* sort_ws = embed_sort_supported(&e) ? (const char*)a->scratch + sl.o_sort : nullptr;
  return embed_ln_bwd_ex(&e, a->dx, a->d_word_emb, a->d_pos_emb, a->d_type_emb, a->d_emb_ln_g, a->d_emb_ln_b, s.part, s.dy, (hipStream_t)stream, sort_ws);
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 DEBUG forward (dropout off): the graph of carel_encoder_forward with every stored value in fp32 -- linears on the
// f32-input matrix cores (carel_sgemm_f32: exact fp32 products, fp32 accumulation), attention, GELU (erff) and LayerNorm in
// fp32 -- so that a difference from the fp32 reference is kernel error, not bf16 rounding.  Measurement / test tool:
// ~20x slower than the bf16 path, forward only, dense rows only.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

// one wave per (sample, head, query): scores over the S <= 128 keys, softmax, P V.  qkv f32 [T, 2304] = q | k | v blocks of 768.
__global__ __launch_bounds__(64) void attn_f32_kernel(const float* __restrict__ qkv, const long* __restrict__ mask, const float* __restrict__ rel,
                                                      float* __restrict__ ctx, int S) {
  const int lane = threadIdx.x;
  const int q = blockIdx.x % S, h = (blockIdx.x / S) % ENH, b = blockIdx.x / (S * ENH);
  __shared__ float qs[64], ps[128];
  const float* base = qkv + (long)b * S * (3 * EH);
  qs[lane] = base[(long)q * (3 * EH) + h * 64 + lane];
  __syncthreads();
  float sc[2], mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = lane + i * 64;
    sc[i] = -INFINITY;
    if (k < S && (!mask || mask[(long)b * S + k] != 0)) {
      const float* kr = base + (long)k * (3 * EH) + EH + h * 64;
      float d = 0.f;
      for (int e = 0; e < 64; ++e) d = fmaf(qs[e], kr[e], d);
      d *= 0.125f;                                             // 1 / sqrt(64)
      if (rel) d += rel[h * 256 + 127 + k - q];
      sc[i] = d;
    }
    mx = fmaxf(mx, sc[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float pr[2], sum = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = lane + i * 64;
    // a sample with no attended key (the batch's filler samples): HF's additive finfo.min mask gives the uniform distribution
    pr[i] = k < S ? (mx == -INFINITY ? 1.f : expf(sc[i] - mx)) : 0.f;
    sum += pr[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  ps[lane] = pr[0] / sum; ps[lane + 64] = pr[1] / sum;
  __syncthreads();
  float acc = 0.f;
  for (int k = 0; k < S; ++k) acc = fmaf(ps[k], base[(long)k * (3 * EH) + 2 * EH + h * 64 + lane], acc);
  ctx[((long)b * S + q) * EH + h * 64 + lane] = acc;
}

__global__ __launch_bounds__(256) void gelu_f32_kernel(float* __restrict__ u, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float x = u[i]; u[i] = 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
}

struct F32Work { float *x, *x1, *h, *ctx, *qkv, *u, *stats; char* xb; size_t total; };
F32Work f32_work(char* base, long B, long S) {
  const size_t T = (size_t)B * S;
  F32Work w; size_t o = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += al(bytes); return p; };
  w.x = (float*)take(T * EH * 4); w.x1 = (float*)take(T * EH * 4); w.h = (float*)take(T * EH * 4); w.ctx = (float*)take(T * EH * 4);
  w.qkv = (float*)take(T * 3 * EH * 4); w.u = (float*)take(T * EI * 4); w.stats = (float*)take(T * 2 * 4); w.xb = take(T * EH * 2);
  w.total = o;
  return w;
}

}  // namespace

extern "C" int64_t carel_encoder_f32_work_bytes(int32_t batch, int32_t seq_len) {
  return (int64_t)f32_work(nullptr, batch, seq_len).total;
}

extern "C" int carel_encoder_forward_f32(const carel_encoder_args* a, void* work, void* x_out, void* stream_) {
  const char* who = "carel_encoder_forward_f32";
  if (!a || !work || !x_out) return set_error(CAREL_ERR_ARG, "%s: null argument", who);
  if (a->hidden != EH || a->heads != ENH || a->intermediate != EI) return set_error(CAREL_ERR_SHAPE, "%s: only the BERT-base geometry (768/12/3072) is supported", who);
  if (a->batch < 1 || a->n_layers < 1 || a->seq_len < 1 || a->seq_len > 128) return set_error(CAREL_ERR_SHAPE, "%s: bad batch / n_layers / seq_len (<= 128)", who);
  if (a->tok_row || a->n_tokens || a->cu_seqlens) return set_error(CAREL_ERR_ARG, "%s: dense batches only (no token packing)", who);
  if (!a->input_ids || !a->layers || !a->word_emb || !a->pos_emb || !a->type_emb || !a->emb_ln_g || !a->emb_ln_b) return set_error(CAREL_ERR_ARG, "%s: null tensor", who);
  hipStream_t stream = (hipStream_t)stream_;
  const long B = a->batch, S = a->seq_len, T = B * S;
  const F32Work w = f32_work((char*)work, B, S);
  int rc;
  carel_embed_args e;
  e.input_ids = a->input_ids; e.token_type_ids = a->token_type_ids; e.word_emb = a->word_emb; e.pos_emb = a->pos_emb; e.type_emb = a->type_emb;
  e.ln_gamma = a->emb_ln_g; e.ln_beta = a->emb_ln_b; e.ln_eps = a->ln_eps; e.batch = a->batch; e.seq_len = a->seq_len; e.hidden = EH;
  e.vocab_size = a->vocab_size; e.max_pos = a->max_pos; e.type_vocab = a->type_vocab; e.roberta = a->roberta; e.pad_id = a->pad_id;
  e.drop_seed = 0; e.drop_idx_offset = 0; e.drop_p = 0.f; e.x_f32 = w.x; e.x_bf16 = w.xb; e.stats = w.stats; e.tok_row = nullptr; e.n_rows = 0;
  if ((rc = carel_embed_ln_fwd(&e, stream))) return rc;
  const size_t row_bytes = (size_t)T * EH * 4;
  for (int i = 0; i < a->n_layers; ++i) {
    const carel_layer_params& p = a->layers[i];                 // every weight f32 here
    float* xin = w.x;
    float* xout = i + 1 == a->n_layers ? (float*)x_out : w.x;
    if ((rc = carel_sgemm_f32(xin, EH, 0, p.qkv_w, EH, 0, w.qkv, 3 * EH, (int)T, 3 * EH, EH, p.qkv_b, 0, 1, 0, stream))) return rc;
    hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)(B * ENH * S)), dim3(64), 0, stream, (const float*)w.qkv, (const long*)a->attention_mask,
                       (const float*)a->rel_bias_dist, w.ctx, (int)S);
    if ((rc = check_launch("attn_f32_kernel"))) return rc;
    if (hipMemcpyAsync(w.h, xin, row_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return set_error(CAREL_ERR_HIP, "%s: copy failed", who);
    if ((rc = carel_sgemm_f32(w.ctx, EH, 0, p.out_w, EH, 0, w.h, EH, (int)T, EH, EH, p.out_b, 1, 1, 0, stream))) return rc;
    if ((rc = carel_layernorm_fwd(w.h, p.ln1_g, p.ln1_b, a->ln_eps, T, EH, w.x1, nullptr, nullptr, stream))) return rc;
    if ((rc = carel_sgemm_f32(w.x1, EH, 0, p.ffn1_w, EH, 0, w.u, EI, (int)T, EI, EH, p.ffn1_b, 0, 1, 0, stream))) return rc;
    const long nu = T * EI;
    hipLaunchKernelGGL(gelu_f32_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, stream, w.u, nu);
    if ((rc = check_launch("gelu_f32_kernel"))) return rc;
    if (hipMemcpyAsync(w.h, w.x1, row_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return set_error(CAREL_ERR_HIP, "%s: copy failed", who);
    if ((rc = carel_sgemm_f32(w.u, EI, 0, p.ffn2_w, EI, 0, w.h, EH, (int)T, EH, EI, p.ffn2_b, 1, 1, 0, stream))) return rc;
    if ((rc = carel_layernorm_fwd(w.h, p.ln2_g, p.ln2_b, a->ln_eps, T, EH, xout, nullptr, nullptr, stream))) return rc;
  }
  return CAREL_OK;
}
