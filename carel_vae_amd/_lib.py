"""ctypes binding of libcarel_hip.so (C ABI declared in include/carel_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is
raised -- nothing here (or anywhere in this package) routes through a CPU/eager implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CAREL_HIP_LIB") or os.path.join(_HERE, "libcarel_hip.so")     # CAREL_HIP_LIB: an experiment build (tools/ablate_*.sh, tools/ab_lib.sh)
ABI_VERSION = 7


class CarelError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("form", C.c_int32), ("epilogue", C.c_int32), ("splits", C.c_int32),
                ("out_bf16", C.c_void_p), ("out2_bf16", C.c_void_p), ("out_f32", C.c_void_p),
                ("bias", C.c_void_p), ("resid_f32", C.c_void_p), ("aux_bf16", C.c_void_p),
                ("drop_seed", C.c_uint32), ("drop_site", C.c_uint32), ("drop_idx_offset", C.c_uint32),
                ("drop_p", C.c_float), ("drop_row_map", C.c_void_p), ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64),
                ("colsum_a", C.c_void_p), ("colsum_part", C.c_void_p), ("splitk_ws_zeroed", C.c_int32),
                ("resid_ln_stats", C.c_void_p), ("resid_ln_gamma", C.c_void_p), ("resid_ln_beta", C.c_void_p)]


class GemmRowLnArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("lda", C.c_int64), ("ldb", C.c_int64), ("M", C.c_int32), ("K", C.c_int32),
                ("bias", C.c_void_p), ("resid_f32", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
                ("h_f32", C.c_void_p), ("x_f32", C.c_void_p), ("x_bf16", C.c_void_p), ("stats", C.c_void_p),
                ("drop_seed", C.c_uint32), ("drop_site", C.c_uint32), ("drop_idx_offset", C.c_uint32), ("drop_p", C.c_float),
                ("drop_row_map", C.c_void_p), ("w_packed", C.c_int32)]


class MmdArgs(C.Structure):
    _fields_ = [("s1", C.c_void_p), ("s2", C.c_void_p), ("ld1", C.c_int64), ("ld2", C.c_int64),
                ("n1", C.c_int32), ("n2", C.c_int32), ("d", C.c_int32), ("n_alphas", C.c_int32),
                ("alphas", C.c_float * 8), ("eps", C.c_float),
                ("mmd_out", C.c_void_p), ("kernels_out", C.c_void_p),
                ("grad_mmd", C.c_void_p), ("g1", C.c_void_p), ("g2", C.c_void_p)]


class PdistArgs(C.Structure):
    _fields_ = [("s1", C.c_void_p), ("s2", C.c_void_p), ("ld1", C.c_int64), ("ld2", C.c_int64),
                ("n1", C.c_int32), ("n2", C.c_int32), ("d", C.c_int32), ("eps", C.c_float),
                ("dist_out", C.c_void_p), ("grad_dist", C.c_void_p), ("g1", C.c_void_p), ("g2", C.c_void_p)]


class EmbedArgs(C.Structure):
    _fields_ = [("input_ids", C.c_void_p), ("token_type_ids", C.c_void_p),
                ("word_emb", C.c_void_p), ("pos_emb", C.c_void_p), ("type_emb", C.c_void_p),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
                ("batch", C.c_int32), ("seq_len", C.c_int32), ("hidden", C.c_int32),
                ("vocab_size", C.c_int32), ("max_pos", C.c_int32), ("type_vocab", C.c_int32),
                ("roberta", C.c_int32), ("pad_id", C.c_int32),
                ("drop_seed", C.c_uint32), ("drop_idx_offset", C.c_uint32), ("drop_p", C.c_float),
                ("x_f32", C.c_void_p), ("x_bf16", C.c_void_p), ("stats", C.c_void_p),
                ("tok_row", C.c_void_p), ("n_rows", C.c_int32)]


class AttnArgs(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("attention_mask", C.c_void_p), ("ctx", C.c_void_p), ("lse", C.c_void_p),
                ("dctx", C.c_void_p), ("dqkv", C.c_void_p),
                ("batch", C.c_int32), ("seq_len", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
                ("drop_seed", C.c_uint32), ("drop_site", C.c_uint32), ("drop_idx_offset", C.c_uint32),
                ("drop_p", C.c_float), ("cu_seqlens", C.c_void_p), ("rel_bias_dist", C.c_void_p), ("d_rel_bias_dist", C.c_void_p),
                ("q_rows", C.c_int32)]


class TailArgs(C.Structure):
    _fields_ = [("batch", C.c_int32), ("seq_len", C.c_int32), ("hidden", C.c_int32), ("ec_dim", C.c_int32),
                ("e_classes", C.c_int32), ("bow_dim", C.c_int32),
                ("x_last_f32", C.c_void_p), ("pooler_w", C.c_void_p), ("pooler_b", C.c_void_p),
                ("head_w", C.c_void_p * 4), ("head_b", C.c_void_p * 4),
                ("emo_w", C.c_void_p), ("emo_b", C.c_void_p), ("cau_w", C.c_void_p), ("cau_b", C.c_void_p),
                ("pair_w", C.c_void_p), ("pair_b", C.c_void_p), ("dec_w", C.c_void_p), ("dec_b", C.c_void_p),
                ("emo_labels", C.c_void_p), ("cau_labels", C.c_void_p), ("pair_labels", C.c_void_p),
                ("bow", C.c_void_p), ("eps_e", C.c_void_p), ("eps_c", C.c_void_p),
                ("w_mmd", C.c_float), ("w_emo", C.c_float), ("w_cau", C.c_float), ("w_pair", C.c_float),
                ("kl_weight", C.c_float), ("label_smoothing", C.c_float),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint32), ("drop_row_offset", C.c_uint32),
                ("mmd_alpha", C.c_float), ("mmd_eps", C.c_float), ("dis_mode", C.c_int32), ("emo_bce", C.c_int32),
                ("global_label_sum", C.c_void_p), ("global_n", C.c_int32), ("global_row_offset", C.c_int32),
                ("z_global", C.c_void_p), ("mmd_grad_scale", C.c_float), ("global_rank_stride", C.c_int32),
                ("global_label_ranks", C.c_int32),
                ("pooled", C.c_void_p), ("lat", C.c_void_p), ("z", C.c_void_p), ("terms", C.c_void_p),
                ("work", C.c_void_p),
                ("d_emo_w", C.c_void_p), ("d_emo_b", C.c_void_p), ("d_cau_w", C.c_void_p), ("d_cau_b", C.c_void_p),
                ("d_pair_w", C.c_void_p), ("d_pair_b", C.c_void_p), ("d_dec_w", C.c_void_p), ("d_dec_b", C.c_void_p),
                ("d_head_w", C.c_void_p * 4), ("d_head_b", C.c_void_p * 4),
                ("d_pooler_w", C.c_void_p), ("d_pooler_b", C.c_void_p), ("dx_last_f32", C.c_void_p),
                ("cls_rows", C.c_void_p), ("n_rows", C.c_int32), ("serial", C.c_int32)]


class AdamArgs(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("shadow_bf16", C.c_void_p), ("n", C.c_int64), ("step", C.c_int64),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("grad_scale", C.c_float), ("skip_lo", C.c_int64), ("skip_hi", C.c_int64), ("skip_flag", C.c_void_p),
                ("grad_scale_dev", C.c_void_p), ("weight_decay", C.c_float), ("decay_segments", C.c_void_p), ("n_decay_segments", C.c_int32),
                ("skip_count", C.c_void_p)]


class LayerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qkv_w", "qkv_b", "out_w", "out_b", "ln1_g", "ln1_b", "ffn1_w", "ffn1_b",
                                          "ffn2_w", "ffn2_b", "ln2_g", "ln2_b")]


class LayerGrads(C.Structure):
    _fields_ = LayerParams._fields_


class EncoderArgs(C.Structure):
    _fields_ = [("batch", C.c_int32), ("seq_len", C.c_int32), ("n_layers", C.c_int32), ("hidden", C.c_int32),
                ("heads", C.c_int32), ("intermediate", C.c_int32),
                ("vocab_size", C.c_int32), ("max_pos", C.c_int32), ("type_vocab", C.c_int32), ("roberta", C.c_int32),
                ("pad_id", C.c_int32), ("inference", C.c_int32),
                ("ln_eps", C.c_float), ("hidden_dropout", C.c_float), ("attn_dropout", C.c_float),
                ("drop_seed", C.c_uint32), ("drop_row_offset", C.c_uint32),
                ("input_ids", C.c_void_p), ("attention_mask", C.c_void_p), ("token_type_ids", C.c_void_p),
                ("word_emb", C.c_void_p), ("pos_emb", C.c_void_p), ("type_emb", C.c_void_p),
                ("emb_ln_g", C.c_void_p), ("emb_ln_b", C.c_void_p),
                ("layers", C.POINTER(LayerParams)), ("act", C.c_void_p), ("scratch", C.c_void_p),
                ("n_tokens", C.c_int32), ("tok_row", C.c_void_p), ("cu_seqlens", C.c_void_p),
                ("n_cls", C.c_int32), ("cls_rows", C.c_void_p), ("cls_orig_rows", C.c_void_p),
                ("overlap_wgrad", C.c_int32),
                ("layer_grads", C.POINTER(LayerGrads)),
                ("d_word_emb", C.c_void_p), ("d_pos_emb", C.c_void_p), ("d_type_emb", C.c_void_p),
                ("d_emb_ln_g", C.c_void_p), ("d_emb_ln_b", C.c_void_p), ("dx", C.c_void_p),
                ("rel_bias_dist", C.c_void_p), ("d_rel_bias_dist", C.c_void_p)]


class HostPackArgs(C.Structure):
    _fields_ = [("input_ids", C.c_void_p), ("attention_masks", C.c_void_p), ("token_type_ids", C.c_void_p), ("labels", C.c_void_p),
                ("cau_labels", C.c_void_p), ("emo_labels", C.c_void_p), ("bow_cols", C.c_void_p), ("bow_vals", C.c_void_p),
                ("idx", C.c_void_p), ("dst", C.c_void_p), ("n_samples", C.c_int64), ("batch", C.c_int32), ("seq_len", C.c_int32),
                ("bow_entries", C.c_int32), ("emo_is_float", C.c_int32), ("off_input_ids", C.c_int64), ("off_attention_masks", C.c_int64),
                ("off_token_type_ids", C.c_int64), ("off_labels", C.c_int64), ("off_cau_labels", C.c_int64), ("off_emo_labels", C.c_int64),
                ("off_trip", C.c_int64), ("lengths", C.c_void_p), ("batch_padded", C.c_int32), ("off_cu", C.c_int64), ("off_tok", C.c_int64),
                ("t_eff", C.c_int64), ("t_pad", C.c_int64)]


class HsicArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("ldx", C.c_int64), ("ldy", C.c_int64), ("m", C.c_int32), ("d", C.c_int32),
                ("s_x", C.c_float), ("s_y", C.c_float), ("hsic_out", C.c_void_p), ("grad_hsic", C.c_void_p),
                ("gx", C.c_void_p), ("gy", C.c_void_p)]


class ViArgs(C.Structure):
    _fields_ = [("z", C.c_void_p), ("batch", C.c_int32), ("ec_dim", C.c_int32), ("net", C.c_void_p * 8), ("perm", C.c_void_p),
                ("loss_out", C.c_void_p), ("d_net", C.c_void_p * 8), ("dz", C.c_void_p)]


class EnTailArgs(C.Structure):
    """carel_en_tail_args (include/carel_hip.h)."""
    _fields_ = [("batch", C.c_int32), ("seq_len", C.c_int32), ("hidden", C.c_int32), ("ec_dim", C.c_int32), ("con_dim", C.c_int32),
                ("bow_dim", C.c_int32),
                ("x_last_f32", C.c_void_p), ("cls_rows", C.c_void_p), ("n_rows", C.c_int32),
                ("pooler_w", C.c_void_p), ("pooler_b", C.c_void_p),
                ("head_w", C.c_void_p * 6), ("head_b", C.c_void_p * 6),
                ("cdisc_w", C.c_void_p), ("cdisc_b", C.c_void_p),
                ("sdisc_w", C.c_void_p * 4), ("sdisc_b", C.c_void_p * 4),
                ("ccls_w", C.c_void_p), ("ccls_b", C.c_void_p),
                ("emo_w", C.c_void_p), ("emo_b", C.c_void_p), ("cau_w", C.c_void_p), ("cau_b", C.c_void_p),
                ("pair_w", C.c_void_p), ("pair_b", C.c_void_p), ("dec_w", C.c_void_p), ("dec_b", C.c_void_p),
                ("emo_labels", C.c_void_p), ("cau_labels", C.c_void_p), ("pair_labels", C.c_void_p),
                ("bow", C.c_void_p), ("eps", C.c_void_p),
                ("w_con_adv", C.c_float), ("w_ec_adv", C.c_float), ("w_ecce_adv", C.c_float), ("w_ec_mul", C.c_float),
                ("w_con_mul", C.c_float), ("w_pair", C.c_float), ("kl_w_ec", C.c_float), ("kl_w_con", C.c_float),
                ("label_smoothing", C.c_float), ("epsilon", C.c_float), ("drop_p", C.c_float), ("drop_seed", C.c_uint32),
                ("drop_row_offset", C.c_uint32), ("global_label_sum", C.c_void_p), ("global_n", C.c_int32),
                ("pooled", C.c_void_p), ("lat", C.c_void_p), ("z", C.c_void_p), ("terms", C.c_void_p), ("work", C.c_void_p),
                ("g_cdisc_w", C.c_void_p * 3), ("g_cdisc_b", C.c_void_p * 3),
                ("g_sdisc_w", C.c_void_p * 4), ("g_sdisc_b", C.c_void_p * 4),
                ("g_sdisc_ent_w", C.c_void_p * 4), ("g_sdisc_ent_b", C.c_void_p * 4),
                ("d_ccls_w", C.c_void_p), ("d_ccls_b", C.c_void_p), ("d_emo_w", C.c_void_p), ("d_emo_b", C.c_void_p),
                ("d_cau_w", C.c_void_p), ("d_cau_b", C.c_void_p), ("d_pair_w", C.c_void_p), ("d_pair_b", C.c_void_p),
                ("d_dec_w", C.c_void_p), ("d_dec_b", C.c_void_p), ("d_pooler_w", C.c_void_p), ("d_pooler_b", C.c_void_p),
                ("dx_last_f32", C.c_void_p)]


GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS_BF16, EPI_BIAS_GELU, EPI_BIAS_DROP_RESID, EPI_DGELU_BF16, EPI_ADD_F32, EPI_SLAB_F32, EPI_BIAS_GELU_DG, EPI_MUL_BF16 = range(8)

# name -> (restype, argtypes); kept in one table so tests can check every symbol of the header exports
class WgradProblem(C.Structure):
    _fields_ = [("dY", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32)]


class LnPartialSet(C.Structure):
    _fields_ = [("partials", C.c_void_p), ("rows", C.c_int64), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dbias", C.c_void_p)]


class WgradGroupArgs(C.Structure):
    _fields_ = [("prob", WgradProblem * 4), ("n_prob", C.c_int32), ("T", C.c_int64), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("ln", LnPartialSet * 2), ("n_ln", C.c_int32)]


SIGNATURES = {
    "carel_abi_version": (C.c_int, []),
    "carel_init": (C.c_int, [C.c_int]),
    "carel_last_error": (C.c_char_p, []),
    "carel_gemm_bf16": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "carel_gemm_wgrad_splits": (C.c_int32, [C.c_int32, C.c_int32, C.c_int64]),
    "carel_profile_gemm": (C.c_int, [C.c_int32, C.c_int32]),
    "carel_profile_gemm_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "carel_profile_gemm_overheads": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "carel_slab_reduce_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "carel_gemm_wgrad_group_ws_bytes": (C.c_int64, [C.POINTER(WgradGroupArgs)]),
    "carel_gemm_wgrad_group": (C.c_int, [C.POINTER(WgradGroupArgs), C.c_void_p]),
    "carel_mean_pool_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "carel_mean_pool_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "carel_triplet_semihard": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "carel_grad_norm_clip": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "carel_l2_normalize_fwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "carel_l2_normalize_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "carel_rbf_mmd_fwd": (C.c_int, [C.POINTER(MmdArgs), C.c_void_p]),
    "carel_rbf_mmd_bwd": (C.c_int, [C.POINTER(MmdArgs), C.c_void_p]),
    "carel_pdist_fwd": (C.c_int, [C.POINTER(PdistArgs), C.c_void_p]),
    "carel_pdist_bwd": (C.c_int, [C.POINTER(PdistArgs), C.c_void_p]),
    "carel_hsic_fwd": (C.c_int, [C.POINTER(HsicArgs), C.c_void_p]),
    "carel_hsic_bwd": (C.c_int, [C.POINTER(HsicArgs), C.c_void_p]),
    "carel_selftest_layouts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "carel_embed_ln_fwd": (C.c_int, [C.POINTER(EmbedArgs), C.c_void_p]),
    "carel_embed_ln_bwd_blocks": (C.c_int, [C.c_int64]),
    "carel_embed_ln_bwd": (C.c_int, [C.POINTER(EmbedArgs)] + [C.c_void_p] * 8),
    "carel_layernorm_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "carel_layernorm_bwd_blocks": (C.c_int, [C.c_int64]),
    "carel_layernorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                      C.c_uint32, C.c_uint32, C.c_uint32, C.c_float] + [C.c_void_p] * 7),
    "carel_partial_reduce_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "carel_layernorm_bwd_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                             C.c_uint32, C.c_uint32, C.c_uint32, C.c_float] + [C.c_void_p] * 8),
    "carel_colsum_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int32,
                                    C.c_void_p, C.c_void_p]),
    "carel_tail_workspace_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "carel_tail_latents": (C.c_int, [C.POINTER(TailArgs), C.c_void_p]),
    "carel_tail_losses": (C.c_int, [C.POINTER(TailArgs), C.c_void_p]),
    "carel_tail_backward": (C.c_int, [C.POINTER(TailArgs), C.c_void_p, C.c_void_p]),
    "carel_tail_profile": (C.c_int, [C.c_void_p]),
    "carel_side_stream": (C.c_void_p, [C.c_int32]),
    "carel_encoder_backward_join": (C.c_int, [C.POINTER(EncoderArgs), C.c_void_p]),
    "carel_tail_backward_dz": (C.c_int, [C.POINTER(TailArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "carel_vi_aprx": (C.c_int, [C.POINTER(ViArgs), C.c_void_p]),
    "carel_vi_upper": (C.c_int, [C.POINTER(ViArgs), C.c_void_p]),
    "carel_scale_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "carel_tail_pair_dead_offset": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "carel_pair_probs": (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "carel_adam_step": (C.c_int, [C.POINTER(AdamArgs), C.c_void_p]),
    "carel_cast_f32_to_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "carel_rmsprop_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "carel_encoder_act_bytes": (C.c_int64, [C.c_int32] * 4),
    "carel_encoder_scratch_bytes": (C.c_int64, [C.c_int32] * 2),
    "carel_encoder_x_last": (C.c_void_p, [C.POINTER(EncoderArgs)]),
    "carel_encoder_f32_work_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "carel_encoder_forward_f32": (C.c_int, [C.POINTER(EncoderArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "carel_encoder_forward": (C.c_int, [C.POINTER(EncoderArgs), C.c_void_p]),
    "carel_encoder_backward_layer": (C.c_int, [C.POINTER(EncoderArgs), C.c_int32, C.c_void_p]),
    "carel_encoder_backward_embeddings": (C.c_int, [C.POINTER(EncoderArgs), C.c_void_p]),
    "carel_en_tail_workspace_floats": (C.c_int64, [C.c_int32] * 4),
    "carel_en_tail_latents": (C.c_int, [C.POINTER(EnTailArgs), C.c_void_p]),
    "carel_en_tail_losses": (C.c_int, [C.POINTER(EnTailArgs), C.c_void_p]),
    "carel_en_tail_backward": (C.c_int, [C.POINTER(EnTailArgs), C.c_void_p, C.c_void_p]),
    "carel_en_pair_logits": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "carel_host_pack_batch": (C.c_int, [C.POINTER(HostPackArgs)]),
    "carel_bow_expand": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "carel_axpy_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "carel_sgemm_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "carel_attention_fwd": (C.c_int, [C.POINTER(AttnArgs), C.c_void_p]),
    "carel_attention_bwd": (C.c_int, [C.POINTER(AttnArgs), C.c_void_p]),
    "carel_relpos_expand": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "carel_relpos_reduce": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
}

# entry points of the EXPERIMENTS build only (include/carel_hip_experiments.h): tuning hooks + the kernels that were not adopted
EXP_SIGNATURES = {
    "carel_gemm_set_variant": (C.c_int, [C.c_int32]),
    "carel_gemm_rowln": (C.c_int, [C.POINTER(GemmRowLnArgs), C.c_void_p]),
    "carel_gemm_rowln_pack": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
}
EXP_LIB_PATH = os.path.join(_HERE, "libcarel_hip_exp.so")

_lib = None          # the ACTIVE library: the product library unless an experiments() block is open
_product = None
_exp = None


def _open(path, sigs, what):
    # torch ships its own HIP runtime (libamdhip64); it must be the one already loaded when our library's
    # dependency is resolved, otherwise two runtimes coexist and torch sees no GPU.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise CarelError(
            "%s not found at %s -- build it with `python -m carel_vae_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback." % (what, path))
    lib = C.CDLL(path)
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    v = lib.carel_abi_version()
    if v != ABI_VERSION:
        raise CarelError("%s ABI %d != binding ABI %d: rebuild" % (what, v, ABI_VERSION))
    return lib


def load():
    """The active library (loaded once); raises loudly if it is absent or of the wrong ABI."""
    global _lib, _product
    if _lib is not None:
        return _lib
    if os.environ.get("CAREL_USE_EXPERIMENTS") == "1":       # A/B tools (tools/*.py that flip hooks): the whole process on the experiments build
        _lib = load_experiments()
        return _lib
    _product = _open(LIB_PATH, SIGNATURES, "libcarel_hip.so")
    _lib = _product
    return _lib


def load_experiments():
    """libcarel_hip_exp.so (every product entry point + EXP_SIGNATURES), without making it the active library."""
    global _exp
    if _exp is None:
        sigs = dict(SIGNATURES)
        sigs.update(EXP_SIGNATURES)
        _exp = _open(os.environ.get("CAREL_HIP_EXP_LIB") or EXP_LIB_PATH, sigs, "libcarel_hip_exp.so")
    return _exp


class experiments:
    """`with _lib.experiments():` -- inside the block load() returns the EXPERIMENTS library, so that everything (the model, ops, the
    tests' direct calls) runs on the build that has carel_gemm_set_variant and the non-adopted kernels.  Tests and A/B tools only; the
    two libraries keep separate side streams / events / GELU tables, so do not mix objects created under one with calls under the other."""

    def __enter__(self):
        global _lib
        load()
        self._prev = _lib
        _lib = load_experiments()
        return _lib

    def __exit__(self, *a):
        global _lib
        _lib = self._prev


def check(rc, what=""):
    if rc != 0:
        msg = load().carel_last_error().decode("utf8", "replace")
        raise CarelError("%s failed (%d): %s" % (what or "carel call", rc, msg))


_INITED = set()


def ensure_init(device=None):
    """carel_init(device) once per device per process and library (the library's per-device immutable state: the GELU table)."""
    import torch
    d = torch.cuda.current_device() if device is None else int(device)
    lib = load()
    if (id(lib), d) not in _INITED:
        check(lib.carel_init(d), "carel_init")
        _INITED.add((id(lib), d))


def current_stream():
    import torch
    ensure_init()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """data_ptr of a tensor (or None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())
