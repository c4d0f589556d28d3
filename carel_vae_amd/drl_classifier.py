"""`DrlClassifier`, `MMDStatistic`, `pdist` with the module surface of the reference's
drl_classifier_ec_mmd_final_mul.py (:149-596), executed by the HIP kernels of libcarel_hip.so.

PyTorch's role here is autograd *glue* and memory: one `torch.autograd.Function` spans the whole training
forward (encoder + VAE tail -> scalar loss); its backward enqueues the HIP backward and leaves the
gradients in a flat fp32 buffer that every `Parameter.grad` aliases.  There is no eager/CPU fallback:
calling the model with CPU tensors raises.
"""
import ctypes as C
import math
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops

H, I_FF, NH = 768, 3072, 12

DEFAULT_OPT = dict(language="zh", max_len=128, e_num_class=6, c_num_class=1, pair_num_class=1, ec_dim=24, bert_dim=768,
                   kl_ann_iterations=20000, epochs=20, batch_size=64, ec_kl_lambda=0.03, label_smoothing=0.1,
                   mmd_loss_weight=30.0, emo_mul_loss_weight=10.0, cau_mul_loss_weight=10.0, pair_mul_loss_weight=30.0,
                   dropout=0.5, epsilon=1e-8, vae_lr=1e-5, aprx_lr=0.003, pair_bow_dim=23771, self_iteration=50, self_epochs=10,
                   self_strategy="random", best_model_path="ECPE_model/best_cause_pair_model", model_id="carel")


def make_opt(**kw):
    """The reference's argparse namespace (:30-61) with its defaults."""
    d = dict(DEFAULT_OPT)
    d.update(kw)
    return SimpleNamespace(**d)


REL_KEY = "encoder.encoder.relative_attention_bias.weight"


def encoder_config(language="zh", **kw):
    """BERT-base geometry of `hfl/chinese-roberta-wwm-ext` (:159) or `roberta-base` (:162); "mpnet": the encoder of
    sentence-transformers/all-mpnet-base-v2 (en_ec_sentence_transformer.py:22; transformers MPNetConfig defaults): RoBERTa-style
    position ids, no token types (type_vocab 1 = an all-zero placeholder row that is never updated) and a learned
    relative-position attention bias [32 buckets, 12 heads] shared by all layers (rel_pos)."""
    if language == "mpnet":
        cfg = dict(vocab_size=30527, max_pos=514, type_vocab=1, ln_eps=1e-5, roberta=1, pad_id=1, rel_pos=True)
    elif language == "en":
        cfg = dict(vocab_size=50265, max_pos=514, type_vocab=1, ln_eps=1e-5, roberta=1, pad_id=1)
    else:
        cfg = dict(vocab_size=21128, max_pos=512, type_vocab=2, ln_eps=1e-12, roberta=0, pad_id=0)
    cfg.setdefault("rel_pos", False)
    cfg.update(layers=12, hidden_dropout=0.1, attn_dropout=0.1)
    cfg.update(kw)
    return SimpleNamespace(**cfg)


# ----------------------------------------------------------------------------------------------
# module tree that reproduces the reference's state_dict key names
# ----------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    """A parameter holder named like the HF / nn.Linear module it stands for (weight [, bias])."""

    def __init__(self, wshape, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(wshape))
        if bias:
            self.bias = nn.Parameter(torch.empty(wshape[0]))


class _Self(nn.Module):
    def __init__(self):
        super().__init__()
        self.query, self.key, self.value = _Holder((H, H)), _Holder((H, H)), _Holder((H, H))


class _SelfOutput(nn.Module):
    def __init__(self, k):
        super().__init__()
        self.dense = _Holder((H, k))
        self.LayerNorm = _Holder((H,))


class _Attention(nn.Module):
    def __init__(self):
        super().__init__()
        self.self = _Self()
        self.output = _SelfOutput(H)


class _Intermediate(nn.Module):
    def __init__(self):
        super().__init__()
        self.dense = _Holder((I_FF, H))


class _Layer(nn.Module):
    def __init__(self):
        super().__init__()
        self.attention = _Attention()
        self.intermediate = _Intermediate()
        self.output = _SelfOutput(I_FF)


class _Stack(nn.Module):
    def __init__(self, n, rel_pos=False):
        super().__init__()
        if rel_pos:      # registered BEFORE the layers: in the flat buffer it sits with the embeddings, whose gradients are the last to complete
            self.relative_attention_bias = _Holder((32, NH), bias=False)
        self.layer = nn.ModuleList([_Layer() for _ in range(n)])


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = _Holder((cfg.vocab_size, H), bias=False)
        self.position_embeddings = _Holder((cfg.max_pos, H), bias=False)
        self.token_type_embeddings = _Holder((cfg.type_vocab, H), bias=False)
        self.LayerNorm = _Holder((H,))


class _Pooler(nn.Module):
    def __init__(self):
        super().__init__()
        self.dense = _Holder((H, H))


class CarelEncoder(nn.Module):
    """Parameter container with the key names of transformers BertModel / RobertaModel."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Stack(cfg.layers, getattr(cfg, "rel_pos", False))
        self.pooler = _Pooler()


def _init_like_reference(model, gen=None):
    """HF init for the encoder (normal(0, 0.02), LayerNorm 1/0, zero biases) and nn.Linear default init
    for the heads; pretrained checkpoints are loaded with load_state_dict (same key names)."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.startswith("encoder."):
                if "LayerNorm.weight" in name:
                    p.fill_(1.0)
                elif name.endswith(".bias"):
                    p.zero_()
                else:
                    p.normal_(0.0, 0.02, generator=gen)
            elif name.endswith(".weight"):
                bound = 1.0 / math.sqrt(p.shape[1])
                p.uniform_(-bound, bound, generator=gen)
            else:
                w = dict(model.named_parameters())[name[:-5] + ".weight"]
                bound = 1.0 / math.sqrt(w.shape[1])
                p.uniform_(-bound, bound, generator=gen)


# ----------------------------------------------------------------------------------------------
# autograd glue
# ----------------------------------------------------------------------------------------------
class _TrainLoss(torch.autograd.Function):
    """Whole-step forward; `anchor` is a dummy requires-grad tensor that makes autograd call backward.
    Gradients are written by the kernels straight into model._flat_grad (aliased by every .grad)."""

    @staticmethod
    def forward(ctx, anchor, model, call):
        ctx.model, ctx.call = model, call
        ctx.set_materialize_grads(False)
        model._run_forward(call, training=True)
        # second output: the sampled embeddings [z_e | z_c]; further loss terms built on them (the CLUB bound of the VI
        # ablation, or any torch expression) send their gradient back through `grad_z`
        return call.buf.terms[8].clone(), call.buf.z.clone()

    @staticmethod
    def backward(ctx, grad_out, grad_z):
        if grad_out is None:
            grad_out = ctx.call.buf.terms.new_zeros(())
        ctx.model._run_backward(ctx.call, grad_out, grad_z)
        return None, None, None


class _AprxLoss(torch.autograd.Function):
    """`get_ec_aprx_loss` of drl_classifier_ec_vi.py:422-427: loss of the approximation network p(e|c) on the sampled
    embeddings; its backward fills ONLY the eight ec_mu / ec_log_var gradients.  (In the reference the term also sends a
    gradient into the encoder through e_embedding, but `vae_and_cls_opt.zero_grad()` (:771) discards it before use.)"""

    @staticmethod
    def forward(ctx, anchor, model, z):
        ctx.model = model
        loss, ctx.grads = ops.vi_aprx(z, model._aprx_weights())
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        m = ctx.model
        for k, gk in zip(m._aprx_names, ctx.grads):
            view = m._grad_view(k)
            p = m._named[k]
            if p.grad is None:
                view.copy_(gk * g)
            else:
                view.add_(gk * g)
            p.grad = view
        return None, None, None


class _UpperLoss(torch.autograd.Function):
    """`get_ec_upper_loss` (:429-440): CLUB upper bound of I(e; c) with the negatives e[randperm]."""

    @staticmethod
    def forward(ctx, e, c, model, perm):
        D = e.shape[1]
        z = torch.cat((e, c), dim=1).float().contiguous()
        loss, dz = ops.vi_upper(z, model._aprx_weights(), perm)
        ctx.save_for_backward(dz)
        ctx.D = D
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dz, = ctx.saved_tensors
        dz = dz * g
        return dz[:, :ctx.D], dz[:, ctx.D:], None, None


class _MMDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s1, s2, alphas, ret_matrix):
        s1c, s2c = s1.contiguous().float(), s2.contiguous().float()
        out, kern = ops.rbf_mmd(s1c, s2c, alphas, ret_matrix=ret_matrix)
        ctx.save_for_backward(s1c, s2c)
        ctx.alphas = alphas
        if ret_matrix:
            ctx.mark_non_differentiable(kern)
            return out.reshape(()), kern
        return out.reshape(())

    @staticmethod
    def backward(ctx, g, *unused):
        s1, s2 = ctx.saved_tensors
        g1, g2 = ops.rbf_mmd_backward(s1, s2, ctx.alphas, g.reshape(1).float())
        return g1, g2, None, None


class _HSICFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, s_x, s_y):
        xc, yc = x.contiguous().float(), y.contiguous().float()
        ctx.save_for_backward(xc, yc)
        ctx.s = (s_x, s_y)
        return ops.hsic(xc, yc, s_x, s_y).reshape(())

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx, gy = ops.hsic_backward(x, y, g, *ctx.s)
        return gx, gy, None, None


def HSIC(x, y, s_x=1, s_y=1):
    """Drop-in for `HSIC` of the ablation script drl_classifier_ec_hsic.py:540-547 (HIP kernel, differentiable)."""
    return _HSICFn.apply(x, y, float(s_x), float(s_y))


class MMDStatistic:
    """Drop-in for the reference's `MMDStatistic` (:537-577); `__call__` runs the HIP RBF-MMD kernel."""

    def __init__(self, n_1, n_2):
        self.n_1, self.n_2 = n_1, n_2
        self.a00 = 1. / (n_1 * (n_1 - 1))
        self.a11 = 1. / (n_2 * (n_2 - 1))
        self.a01 = - 1. / (n_1 * n_2)

    def __call__(self, sample_1, sample_2, alphas, ret_matrix=False):
        if sample_1.shape[0] != self.n_1 or sample_2.shape[0] != self.n_2:
            raise ValueError("sample sizes differ from the constructor's n_1/n_2")
        ops._chk_cuda(sample_1, sample_2)
        return _MMDFn.apply(sample_1, sample_2, [float(a) for a in alphas], bool(ret_matrix))

    def pval(self, distances, n_permutations=1000):   # dead in the reference too (:571-577 calls a stub)
        return permutation_test_mat(distances, self.n_1, self.n_2, n_permutations, a00=self.a00, a11=self.a11, a01=self.a01)


def permutation_test_mat(*args, **kwargs):  # the reference's stub (:599-600)
    pass


class _PdistFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s1, s2, eps):
        s1c, s2c = s1.contiguous().float(), s2.contiguous().float()
        ctx.save_for_backward(s1c, s2c)
        ctx.eps = eps
        return ops.pdist(s1c, s2c, eps)

    @staticmethod
    def backward(ctx, g):
        s1, s2 = ctx.saved_tensors
        g1, g2 = ops.pdist_backward(s1, s2, g, ctx.eps)
        return g1, g2, None


def pdist(sample_1, sample_2, norm=2, eps=1e-5):
    """Reference `pdist` (:580-596), L2 only (the only live branch): sqrt(eps + |d2|) from the squared distance itself
    (HIP kernel `carel_pdist_fwd`, differentiable) -- no exp / log round trip, so samples that are far apart (d2 > 100,
    where exp(-d2) underflows in fp32) and samples that nearly coincide both come out to fp32 rounding."""
    if float(norm) != 2.:
        raise NotImplementedError("only norm=2 is on the hot path (:583)")
    ops._chk_cuda(sample_1, sample_2)
    return _PdistFn.apply(sample_1, sample_2, float(eps))


# ----------------------------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------------------------
class _Call:
    """Everything one forward/backward pair shares."""
    pass


class DrlClassifier(nn.Module):
    """Reference `DrlClassifier` (:149-534).  Same constructor argument, attribute names, `forward`,
    `get_pair_preds`, `get_params`, state_dict keys."""

    TERM_NAMES = ("partial", "mmd", "emo", "cau", "pair", "kl_e", "kl_c", "rec", "loss")

    def __init__(self, opt, encoder_cfg=None, seed=None):
        super().__init__()
        self.opt = opt
        self.cfg = encoder_cfg if encoder_cfg is not None else encoder_config(getattr(opt, "language", "zh"))
        if opt.bert_dim != H:
            raise L.CarelError("bert_dim must be 768 (BERT-base kernels)")
        self.encoder = CarelEncoder(self.cfg)
        self._aprx_names = []
        self._has_pair_skip = True          # the pair head is frozen for a step whose pair loss was replaced by 0 (ref :510-511)
        self.strict_pair_skip = False       # True: also leave its .grad None on such a step (for stock torch optimisers; one host read per step)
        self._build_heads(opt)
        gen = None
        if seed is not None:
            gen = torch.Generator().manual_seed(seed)
        _init_like_reference(self, gen)
        self.dropout_base_seed = 0x5EED
        self.varlen = True                   # skip padded positions (results identical; see _pack_info)
        self.cls_only_last = True            # last layer's row-wise half on the [CLS] rows only (results identical)
        self.overlap_wgrad = True            # weight-gradient GEMMs on a second stream beside the dgrad chain (results identical)
        self.forward_chains = False          # dense batches: the two halves of the batch as two forward chains on two streams (results
                                             # identical; opt-in: ~1 % slower than one chain since the round-2 GEMMs fill the chip)
        self.debug_fp32 = False              # forward-only measurement mode: the encoder in fp32 (carel_encoder_forward_f32), dropout off;
                                             # forward_terms() / forward() under torch.no_grad() only -- separates bf16 rounding from kernel error
        self._fwd_count = 0
        self._noise = None
        self._ws = {}                        # small per-shape buffers (tail, [CLS] index arrays)
        self._enc_ws = {}                    # encoder activation / scratch workspaces, LRU-bounded (see _workspace)
        self._dp = None                     # set by carel_vae_amd.dp.DataParallel
        self._adam_hook = None              # set by FusedAdam(fuse_into_backward=True)
        self._flat = None
        self._flatten()

    def _build_heads(self, opt):
        self.emotion_mu = _Holder((opt.ec_dim, H))
        self.emotion_log_var = _Holder((opt.ec_dim, H))
        self.cause_mu = _Holder((opt.ec_dim, H))
        self.cause_log_var = _Holder((opt.ec_dim, H))
        self.emotion_classifier = _Holder((opt.e_num_class, opt.ec_dim))
        self.cause_classifier = _Holder((opt.c_num_class, opt.ec_dim))
        self.pair_classifier = _Holder((opt.pair_num_class, opt.ec_dim * 2))
        self.decoder = _Holder((opt.pair_bow_dim, opt.ec_dim * 2))
        self.dropout = nn.Dropout(opt.dropout)       # probability holder; the mask is drawn in-kernel
        if getattr(opt, "disentangle", "mmd") == "vi":      # approximation network p(e|c), drl_classifier_ec_vi.py:156-163
            D = opt.ec_dim
            self.ec_mu = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, D))
            self.ec_log_var = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, D), nn.Tanh())
            self._aprx_names = [f"{n}.{i}.{t}" for n in ("ec_mu", "ec_log_var") for i in (0, 2) for t in ("weight", "bias")]

    # ------------------------------------------------------------------ flat parameter storage
    def _param_order(self):
        """Flat layout: optimised tensors first (so fused Adam is one contiguous range), the four latent
        heads (never optimised, ref :292-295) last; q/k/v weights and biases adjacent (fused QKV GEMM)."""
        order, _, named = self._param_order_encoder()
        order += ["encoder.pooler.dense.weight", "encoder.pooler.dense.bias", "decoder.weight", "decoder.bias",
                  "emotion_classifier.weight", "emotion_classifier.bias", "cause_classifier.weight", "cause_classifier.bias"]
        self._pair_range_names = ["pair_classifier.weight", "pair_classifier.bias"]
        order += self._pair_range_names
        n_opt_names = len(order)
        order += ["emotion_mu.weight", "emotion_mu.bias", "emotion_log_var.weight", "emotion_log_var.bias",
                  "cause_mu.weight", "cause_mu.bias", "cause_log_var.weight", "cause_log_var.bias"]
        order += self._aprx_names            # own optimiser (ref ec_vi :873), fp32 only
        assert set(order) == set(named), "parameter inventory mismatch"
        return order, n_opt_names, named

    def _param_order_encoder(self):
        """Embeddings and encoder layers (q/k/v weights and biases adjacent: fused QKV GEMM); (order, None, named)."""
        named = dict(self.named_parameters())
        order = []
        e = "encoder.embeddings."
        order += [e + "word_embeddings.weight", e + "position_embeddings.weight", e + "token_type_embeddings.weight",
                  e + "LayerNorm.weight", e + "LayerNorm.bias"]
        if getattr(self.cfg, "rel_pos", False):
            order.append(REL_KEY)            # with the embeddings: complete only after the LAST layer's backward, like them
        for l in range(self.cfg.layers):
            p = f"encoder.encoder.layer.{l}."
            order += [p + "attention.self.query.weight", p + "attention.self.key.weight", p + "attention.self.value.weight",
                      p + "attention.self.query.bias", p + "attention.self.key.bias", p + "attention.self.value.bias",
                      p + "attention.output.dense.weight", p + "attention.output.dense.bias",
                      p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias",
                      p + "intermediate.dense.weight", p + "intermediate.dense.bias",
                      p + "output.dense.weight", p + "output.dense.bias", p + "output.LayerNorm.weight", p + "output.LayerNorm.bias"]
        return order, None, named

    def _flatten(self):
        order, n_opt_names, named = self._param_order()
        dev = next(iter(named.values())).device
        offs, o = {}, 0
        for i, k in enumerate(order):
            if i == n_opt_names:
                self._n_opt = o
            offs[k] = o
            o += (named[k].numel() + 63) & ~63        # 256-byte aligned segments
        flat = torch.empty(o, device=dev, dtype=torch.float32)
        flat.zero_()
        with torch.no_grad():
            for k in order:
                p = named[k]
                seg = flat[offs[k]:offs[k] + p.numel()].view(p.shape)
                seg.copy_(p.data.to(torch.float32))
                p.data = seg
        self._flat, self._offs, self._order = flat, offs, order
        self._named = named
        self._flat_grad = torch.zeros_like(flat) if dev.type == "cuda" else None
        self._shadow = torch.empty(o, device=dev, dtype=torch.bfloat16) if dev.type == "cuda" else None
        self._shadow_versions = None
        self._grad_views = None
        self._layer_structs = None
        self._ws = {}
        self._enc_ws = {}
        self._pair_lo = offs[self._pair_range_names[0]]
        self._pair_hi = offs[self._pair_range_names[1]] + named[self._pair_range_names[1]].numel()

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._flatten()
        return r

    def load_state_dict(self, state_dict, strict=True, **kw):
        # buffers such as HF's `encoder.embeddings.position_ids` are not parameters here
        sd = {k: v for k, v in state_dict.items() if not k.endswith(("position_ids", "token_type_ids"))}
        with torch.no_grad():
            r = super().load_state_dict(sd, strict=strict, **kw)
        self._shadow_versions = None
        return r

    def get_params(self):
        """Reference order (:292-295): encoder, decoder, emotion / cause / pair classifiers.  The four latent
        heads are deliberately absent (they are never updated in the reference).
        With opt.disentangle == "vi": the pair (ec_aprx_params, other_params) of drl_classifier_ec_vi.py:291-302."""
        other = (list(self.encoder.parameters()) + list(self.decoder.parameters()) +
                 list(self.emotion_classifier.parameters()) + list(self.cause_classifier.parameters()) +
                 list(self.pair_classifier.parameters()))
        if self._aprx_names:
            return list(self.ec_mu.parameters()) + list(self.ec_log_var.parameters()), other
        return other

    def _aprx_weights(self):
        if not self._aprx_names:
            raise L.CarelError("the approximation network exists only with opt.disentangle == 'vi'")
        return [self._named[k].data for k in self._aprx_names]

    def get_ec_aprx_loss(self, e_embedding, c_embedding):
        """drl_classifier_ec_vi.py:422-427 (the cause embedding is detached there; only the network gets a gradient)."""
        z = torch.cat((e_embedding.detach(), c_embedding.detach()), dim=1).float().contiguous()
        if not torch.is_grad_enabled():
            return ops.vi_aprx(z, self._aprx_weights())[0].reshape(())
        return _AprxLoss.apply(self._flat.new_zeros((), requires_grad=True), self, z)

    def get_ec_upper_loss(self, e_embedding, c_embedding, random_index=None):
        """drl_classifier_ec_vi.py:429-440.  random_index (extension): the permutation to use instead of a fresh
        torch.randperm(batch) (drawn on the host generator like the reference)."""
        n = e_embedding.shape[0]
        perm = torch.randperm(n) if random_index is None else random_index
        perm = perm.to(e_embedding.device, torch.int32).contiguous()
        return _UpperLoss.apply(e_embedding, c_embedding, self, perm)

    # ------------------------------------------------------------------ helpers
    def _require_cuda(self):
        if self._flat is None or self._flat.device.type != "cuda":
            raise L.CarelError("DrlClassifier parameters are on %s; call .to('cuda') first -- the HIP kernels are the only "
                               "implementation of the step path (no CPU fallback)." % (None if self._flat is None else self._flat.device))

    def _versions(self):
        aprx = set(self._aprx_names)         # fp32 only: an update of the approximation net never stales the bf16 shadow
        return sum(p._version for k, p in self._named.items() if k not in aprx)

    def _refresh_shadow(self):
        vers = self._versions()
        if self._shadow_versions != vers:
            L.check(L.load().carel_cast_f32_to_bf16(self._flat.data_ptr(), self._shadow.data_ptr(), self._flat.numel(),
                                                    L.current_stream()), "carel_cast_f32_to_bf16")
            self._shadow_versions = vers

    def mark_shadow_fresh(self):
        """Called by the fused optimiser, which rewrites the bf16 shadow itself."""
        self._shadow_versions = self._versions()

    def _w(self, key, bf16=False):
        off = self._offs[key]
        base = self._shadow if bf16 else self._flat
        return base.data_ptr() + off * (2 if bf16 else 4)

    def _g(self, key):
        return self._flat_grad.data_ptr() + self._offs[key] * 4

    def _layer_arrays(self):
        if self._layer_structs is not None:
            return self._layer_structs
        n = self.cfg.layers
        P, G = (L.LayerParams * n)(), (L.LayerGrads * n)()
        for l in range(n):
            p = f"encoder.encoder.layer.{l}."
            m = dict(qkv_w=p + "attention.self.query.weight", qkv_b=p + "attention.self.query.bias",
                     out_w=p + "attention.output.dense.weight", out_b=p + "attention.output.dense.bias",
                     ln1_g=p + "attention.output.LayerNorm.weight", ln1_b=p + "attention.output.LayerNorm.bias",
                     ffn1_w=p + "intermediate.dense.weight", ffn1_b=p + "intermediate.dense.bias",
                     ffn2_w=p + "output.dense.weight", ffn2_b=p + "output.dense.bias",
                     ln2_g=p + "output.LayerNorm.weight", ln2_b=p + "output.LayerNorm.bias")
            for f, k in m.items():
                setattr(P[l], f, self._w(k, bf16=f.endswith("_w")))
                setattr(G[l], f, self._g(k))
        self._layer_structs = (P, G)
        return self._layer_structs

    def _layer_arrays_f32(self):
        """carel_layer_params with every weight pointing into the fp32 master buffer (carel_encoder_forward_f32)."""
        P, _ = self._layer_arrays()
        n = self.cfg.layers
        F = (L.LayerParams * n)()
        for l in range(n):
            for f, _t in L.LayerParams._fields_:
                setattr(F[l], f, getattr(P[l], f))
            p = f"encoder.encoder.layer.{l}."
            for f, k in dict(qkv_w="attention.self.query.weight", out_w="attention.output.dense.weight", ffn1_w="intermediate.dense.weight",
                             ffn2_w="output.dense.weight").items():
                setattr(F[l], f, self._w(p + k))
        return F

    def _run_encoder_fp32(self, c):
        """The fp32 debug forward: returns the final hidden states f32 [Bp*S, 768] (dense rows)."""
        lib = L.load()
        dev = self._flat.device
        key = ("f32", c.Bp, c.S)
        buf = self._ws.get(key)
        if buf is None:
            buf = SimpleNamespace(work=torch.zeros(lib.carel_encoder_f32_work_bytes(c.Bp, c.S), device=dev, dtype=torch.uint8),
                                  x=torch.zeros((c.Bp * c.S, H), device=dev, dtype=torch.float32),
                                  dummy=SimpleNamespace(act=torch.zeros(256, device=dev, dtype=torch.uint8), scratch=None))
            self._ws[key] = buf
        ea = self._encoder_args(c.ids, c.att, c.tt, buf.dummy, c.Bp, c.S, True, False, 0, 0, None, None)
        F = self._layer_arrays_f32()
        ea.layers = C.cast(F, C.POINTER(L.LayerParams))
        L.check(lib.carel_encoder_forward_f32(C.byref(ea), buf.work.data_ptr(), buf.x.data_ptr(), L.current_stream()), "carel_encoder_forward_f32")
        ea._keep = F
        return ea, buf.x

    def _workspace(self, B, S, inference):
        key = (B, S, bool(inference))
        ws = self._enc_ws.get(key)
        if ws is not None:
            self._enc_ws[key] = self._enc_ws.pop(key)             # most recently used last
        if ws is None:
            # encoder activations / scratch are GBs at B = 64: keep the four most recent (batch, seq, mode) shapes -- e.g. the
            # training batch, the odd last batch and the evaluation chunk -- and drop only the least recently used one
            while len(self._enc_ws) >= 4:
                self._enc_ws.pop(next(iter(self._enc_ws)))
            lib = L.load()
            dev = self._flat.device
            ws = SimpleNamespace()
            # zero-filled once: with token packing, attention tiles read (masked) rows past a sample's last token, and
            # 0 * NaN from never-written memory would poison the MFMA sums; everything written later is finite
            ws.act = torch.zeros(lib.carel_encoder_act_bytes(B, S, self.cfg.layers, int(inference)), device=dev, dtype=torch.uint8)
            ws.scratch = torch.zeros(lib.carel_encoder_scratch_bytes(B, S), device=dev, dtype=torch.uint8)
            self._enc_ws[key] = ws
        return ws

    def _cls_info(self, B, Bp, S, pack, dev):
        """Index arrays of the dead-row elimination (include/carel_hip.h, carel_encoder_args.n_cls); B real samples,
        Bp = batch the encoder runs (padded so that Bp*S is a multiple of 128)."""
        if not self.cls_only_last:
            return None
        n_cls = (Bp + 127) // 128 * 128
        key = ("cls", B, Bp, S)
        base = self._ws.get(key)
        if base is None:
            orig = np.full(n_cls, -1, dtype=np.int32)
            orig[:B] = np.arange(B, dtype=np.int32) * S
            base = SimpleNamespace(orig=torch.from_numpy(orig).to(dev), compact=torch.arange(B, dtype=torch.int32, device=dev))
            self._ws[key] = base
        if pack is None:
            rows = base.orig                      # dense: the [CLS] token of sample i sits at row i*S
        else:
            rows = torch.full((n_cls,), -1, dtype=torch.int32, device=dev)
            rows[:B] = pack.cu[:B]
        return SimpleNamespace(n_cls=n_cls, rows=rows, orig=base.orig, compact=base.compact)

    def _encoder_args(self, ids, att, tt, ws, B, S, inference, train, seed, row_offset, pack=None, cls=None):
        a = L.EncoderArgs()
        if pack is not None:
            a.n_tokens, a.tok_row, a.cu_seqlens = pack.n_tokens, pack.tok_row.data_ptr(), pack.cu.data_ptr()
        if cls is not None:
            a.n_cls, a.cls_rows, a.cls_orig_rows = cls.n_cls, cls.rows.data_ptr(), cls.orig.data_ptr()
        a.overlap_wgrad = (1 if self.overlap_wgrad else 0) | (2 if (self.overlap_wgrad and self.forward_chains) else 0)
        c = self.cfg
        a.batch, a.seq_len, a.n_layers, a.hidden, a.heads, a.intermediate = B, S, c.layers, H, NH, I_FF
        a.vocab_size, a.max_pos, a.type_vocab, a.roberta, a.pad_id, a.inference = (c.vocab_size, c.max_pos, c.type_vocab,
                                                                                      c.roberta, c.pad_id, int(inference))
        a.ln_eps = c.ln_eps
        a.hidden_dropout = c.hidden_dropout if train else 0.0
        a.attn_dropout = c.attn_dropout if train else 0.0
        a.drop_seed, a.drop_row_offset = seed, row_offset
        a.input_ids, a.attention_mask = ids.data_ptr(), (None if att is None else att.data_ptr())
        a.token_type_ids = None if tt is None else tt.data_ptr()
        e = "encoder.embeddings."
        a.word_emb, a.pos_emb, a.type_emb = self._w(e + "word_embeddings.weight"), self._w(e + "position_embeddings.weight"), self._w(e + "token_type_embeddings.weight")
        a.emb_ln_g, a.emb_ln_b = self._w(e + "LayerNorm.weight"), self._w(e + "LayerNorm.bias")
        P, G = self._layer_arrays()
        a.layers = C.cast(P, C.POINTER(L.LayerParams))
        a.layer_grads = C.cast(G, C.POINTER(L.LayerGrads))
        a.act = ws.act.data_ptr()
        a.scratch = None if ws.scratch is None else ws.scratch.data_ptr()
        a.d_word_emb, a.d_pos_emb, a.d_type_emb = self._g(e + "word_embeddings.weight"), self._g(e + "position_embeddings.weight"), self._g(e + "token_type_embeddings.weight")
        a.d_emb_ln_g, a.d_emb_ln_b = self._g(e + "LayerNorm.weight"), self._g(e + "LayerNorm.bias")
        if getattr(c, "rel_pos", False):
            r = self._rel_buffers(int(a.batch))
            L.check(L.load().carel_relpos_expand(self._w(REL_KEY), r.bucket.data_ptr(), r.dist.data_ptr(), L.current_stream()), "carel_relpos_expand")
            a.rel_bias_dist, a.d_rel_bias_dist = r.dist.data_ptr(), r.ddist.data_ptr()
        return a

    def _rel_buffers(self, batch=1):
        """MPNet relative positions: bucket[i] = relative_position_bucket(i - 127) with the expression of transformers
        MPNetEncoder.relative_position_bucket (num_buckets 32, max_distance 128), the bias by distance [12, 256] made from the
        learned table before every forward, and the gradient by distance, one row per (sample, head), that the attention backward
        kernels add into without atomics (bit-reproducible; carel_relpos_reduce sums the samples in order)."""
        r = getattr(self, "_rel", None)
        dev = self._flat.device
        if r is None or r.bucket.device != dev:
            rel = torch.arange(-127, 129, dtype=torch.long)
            n = -rel
            ret = (n < 0).to(torch.long) * 16
            n = torch.abs(n)
            large = 8 + (torch.log(n.float() / 8) / math.log(128 / 8) * 8).to(torch.long)
            ret = ret + torch.where(n < 8, n, torch.min(large, torch.full_like(large, 15)))
            r = self._rel = SimpleNamespace(bucket=ret.to(torch.int32).to(dev).contiguous(),
                                            dist=torch.zeros((NH, 256), device=dev, dtype=torch.float32),
                                            ddist=torch.zeros((NH, 256), device=dev, dtype=torch.float32))
        if r.ddist.shape[0] < batch * NH:
            r.ddist = torch.zeros((batch * NH, 256), device=dev, dtype=torch.float32)
        return r

    @staticmethod
    def _prep_ids(t, Bp):
        t = t.to(torch.long).contiguous()
        if t.shape[0] != Bp:
            pad = torch.zeros((Bp - t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
            t = torch.cat((t, pad), dim=0)
        return t

    @staticmethod
    def _padded_batch(B, S):
        if S < 32 or S > 128 or S % 32:
            raise L.CarelError("sequence length must be 32, 64, 96 or 128 (--max_len); got %d" % S)
        Bp = B
        while (Bp * S) % 128:
            Bp += 1
        return Bp

    def _pack_info(self, att, B, Bp, S, seq_lengths=None):
        """Token packing for padded batches: only attended positions go through the encoder (the reference pads every
        pair to max_len and ~77 % of ECPE tokens are padding).  Exact for prefix-form masks (HF right padding): padded
        positions are never attended to and nothing but [CLS] is read from the last layer, so every output and
        gradient is unchanged; dropout masks and position ids keep using the ORIGINAL (sample, position) index.
        Returns None to run dense (no padding in the batch, non-prefix mask, or varlen disabled)."""
        if not self.varlen or att is None:
            return None
        if seq_lengths is None:
            m = att[:B].to(torch.int32)
            lens = m.sum(1)
            prefix = ((torch.arange(S, device=m.device)[None, :] < lens[:, None]).to(torch.int32) == m).all()
            info = torch.cat((lens, prefix.to(torch.int32).reshape(1))).cpu()        # the one host sync of the packed path
            lens_l, ok = info[:-1].tolist(), bool(info[-1])
        else:
            lens_l, ok = [int(v) for v in seq_lengths], True
        if not ok or min(lens_l) < 1:
            return None
        t_eff = sum(lens_l)
        if t_eff == B * S:
            return None
        if getattr(seq_lengths, "cu", None) is not None and seq_lengths.t_eff == t_eff and seq_lengths.cu.numel() == Bp + 1:
            # carel_vae_amd.data.PrefetchLoader shipped the packing arrays with the batch: nothing to build or copy here
            return SimpleNamespace(n_tokens=seq_lengths.n_tokens, t_eff=t_eff, cu=seq_lengths.cu, tok_row=seq_lengths.tok_row)
        t_pad = (t_eff + 127) // 128 * 128
        lens_a = np.asarray(lens_l, dtype=np.int64)
        cu = np.zeros(Bp + 1, dtype=np.int32)
        cu[1:B + 1] = np.cumsum(lens_a)
        cu[B + 1:] = t_eff
        tok = np.full(t_pad, -1, dtype=np.int32)
        starts = np.repeat(np.arange(B, dtype=np.int64) * S - cu[:B], lens_a)
        tok[:t_eff] = (np.arange(t_eff, dtype=np.int64) + starts).astype(np.int32)
        dev = att.device
        return SimpleNamespace(n_tokens=t_pad, t_eff=t_eff, cu=torch.from_numpy(cu).to(dev, non_blocking=True),
                               tok_row=torch.from_numpy(tok).to(dev, non_blocking=True))

    def set_noise(self, eps_e, eps_c):
        """Test hook: use these two [ec_dim] vectors as the next call's reparameterisation noise (ref :350)."""
        self._noise = None if eps_e is None else (eps_e, eps_c)

    def _draw_noise(self, dev):
        if self._noise is not None:
            e, c = self._noise
            self._noise = None
            return e.to(dev, torch.float32).contiguous(), c.to(dev, torch.float32).contiguous()
        D = self.opt.ec_dim
        if self._dp is not None:                     # same draws on every rank, no collective
            return self._dp.draw_noise(D, dev)
        eps_e = torch.randn(D, device=dev)           # emotion first, then cause (ref :215-216)
        eps_c = torch.randn(D, device=dev)
        return eps_e, eps_c

    # ------------------------------------------------------------------ forward / backward bodies
    def _make_call(self, input_ids, att_masks, token_type_ids, emotion_labels, cause_labels, pair_labels, content_bow, iteration, training,
                   seq_lengths=None):
        self._require_cuda()
        ops._chk_cuda(input_ids, att_masks, token_type_ids, content_bow)
        B, S = input_ids.shape
        Bp = self._padded_batch(B, S)
        c = _Call()
        c.B, c.S, c.Bp = B, S, Bp
        c.ids, c.att = self._prep_ids(input_ids, Bp), self._prep_ids(att_masks, Bp)
        c.tt = None if token_type_ids is None else self._prep_ids(token_type_ids, Bp)
        dev = input_ids.device
        f32 = torch.float32
        c.labels = dict(emo=emotion_labels.to(dev, torch.long).reshape(-1).contiguous(), cau=cause_labels.to(dev, f32).reshape(-1).contiguous(),
                        pair=pair_labels.to(dev, f32).reshape(-1).contiguous(), bow=content_bow.to(dev, f32).contiguous())
        if c.labels["bow"].shape != (B, self.opt.pair_bow_dim):
            raise L.CarelError("content_bow must be [batch, pair_bow_dim]")
        c.iteration = int(iteration)
        c.training = training
        c.eps_e, c.eps_c = self._draw_noise(dev)
        self._fwd_count += 1
        c.seed = (self.dropout_base_seed * 1000003 + self._fwd_count) & 0xFFFFFFFF
        c.row_offset = 0 if self._dp is None else self._dp.row_offset(B)
        c.pack = self._pack_info(c.att, B, Bp, S, seq_lengths)
        key = ("tail", B, S)
        buf = self._ws.get(key)
        if buf is None:
            buf = ops.TailBuffers(B, S, self.opt.ec_dim, self.opt.e_num_class, self.opt.pair_bow_dim, dev, rows=Bp * S)
            self._ws[key] = buf
        c.buf = buf
        if self._dp is not None:
            self._dp.prepare(c)
        return c

    def _tail_weights(self):
        keys = ["encoder.pooler.dense.weight", "encoder.pooler.dense.bias", "decoder.weight", "decoder.bias"]
        for n in ("emotion_mu", "emotion_log_var", "cause_mu", "cause_log_var", "emotion_classifier", "cause_classifier", "pair_classifier"):
            keys += [n + ".weight", n + ".bias"]
        return {k: self._named[k].data for k in keys}, {k: self._grad_view(k) for k in keys}

    def _grad_view(self, k):
        p = self._named[k]
        o = self._offs[k]
        return self._flat_grad[o:o + p.numel()].view(p.shape)

    def _run_forward(self, c, training):
        if self._adam_hook is not None:
            self._adam_hook._join()          # weight updates still in flight on the auxiliary stream
        self._refresh_shadow()
        train_drop = self.training          # dropout follows module mode (model.train() / .eval()), like nn.Dropout
        lib = L.load()
        st = L.current_stream()
        if self.debug_fp32:
            if training:
                raise L.CarelError("debug_fp32 is a forward-only measurement mode: call forward_terms() / forward() under torch.no_grad()")
            train_drop = False
            c.pack, c.cls, ws = None, None, None
            ea, x_f32 = self._run_encoder_fp32(c)
            x_last_ptr = x_f32.data_ptr()
        else:
            ws = self._workspace(c.Bp, c.S, inference=not training)
            c.cls = self._cls_info(c.B, c.Bp, c.S, c.pack, self._flat.device)
            ea = self._encoder_args(c.ids, c.att, c.tt, ws, c.Bp, c.S, not training, train_drop, c.seed, c.row_offset, c.pack, c.cls)
            L.check(lib.carel_encoder_forward(C.byref(ea), st), "carel_encoder_forward")
            x_last_ptr = lib.carel_encoder_x_last(C.byref(ea))
        W, G = self._tail_weights()
        xl = SimpleNamespace(data_ptr=lambda: x_last_ptr)
        klw = ops.kl_anneal_weight(c.iteration, self.opt)
        drop = (self.opt.dropout if train_drop else 0.0, c.seed, c.row_offset)
        if c.cls is not None:           # compact final hidden states: sample i's [CLS] is row i
            cls_rows, n_rows = c.cls.compact, c.cls.n_cls
        else:
            cls_rows, n_rows = (None, c.Bp * c.S) if c.pack is None else (c.pack.cu, c.pack.n_tokens)
        ta = ops.tail_args(c.buf, xl, W, c.labels, c.eps_e, c.eps_c, self.opt, klw, grads=G, drop=drop, cls_rows=cls_rows, n_rows=n_rows)
        ops.tail_latents(ta)
        if self._dp is not None:
            self._dp.fill_global(ta, c)                  # all-gather z, all-reduce label sum
        ta.serial = 0 if self.overlap_wgrad else 1       # side stream allowed: the loss kernel runs beside the decoder passes
        ops.tail_losses(ta)
        c.ea, c.ta, c.ws = ea, ta, ws
        c.keep = (W, G, xl)

    def _run_backward(self, c, grad_out, grad_z=None):
        lib = L.load()
        st = L.current_stream()
        named = self._named
        first = named[self._order[0]]
        accumulate = first.grad is not None
        prev = self._flat_grad.clone() if accumulate else None
        go = grad_out.to(torch.float32).reshape(1).contiguous()      # device scalar, never read on the host
        ea = c.ea
        ea.dx = c.buf.dx_last.data_ptr()          # [Bp*S (or packed n_tokens), 768]; cleared + CLS rows written by the tail backward
        ops.tail_backward(c.ta, go, None if grad_z is None else grad_z.to(torch.float32).contiguous())
        # classifier / decoder gradients were produced for grad_output = 1: one contiguous range of the flat buffer
        lo = self._offs["decoder.weight"]
        ops.scale_(self._flat_grad[lo:self._pair_hi], go)
        if self._dp is not None:
            self._dp.tail_done()
        self._backward_encoder(ea, accumulate)
        if self._dp is not None:
            self._dp.backward_done(None if accumulate else self._adam_hook)
        if accumulate:
            self._flat_grad.add_(prev)
        self._bind_grads()
        if self.strict_pair_skip and self._has_pair_skip and not accumulate:
            # stock optimisers: leave pair_classifier.grad None when the pair loss was replaced by 0, exactly as the reference
            # does (:510-511) -- torch.optim.Adam then skips the parameter.  One device->host read per step (the reference's
            # own `if torch.isinf(...).any()` is such a read too); FusedAdam needs no read (carel_adam_args.skip_flag).
            off = L.load().carel_tail_pair_dead_offset(c.B, self.opt.ec_dim, self.opt.pair_bow_dim)
            if float(c.buf.work[off]) != 0.0:
                for k in self._pair_range_names:
                    self._named[k].grad = None

    def _backward_encoder(self, ea, accumulate):
        """Encoder layers 11..0 and the embeddings, given ea.dx = d loss / d (last hidden states)."""
        lib = L.load()
        st = L.current_stream()

        def layer_ready(l):             # layer l's parameter gradients are complete in the order of the main stream
            work = self._dp.layer_done(l) if self._dp is not None else None
            if self._adam_hook is not None and not accumulate and (self._dp is None or work is not None):
                self._adam_hook._layer_ready(l, after=work)

        rel = self._rel_buffers(int(ea.batch)) if getattr(self.cfg, "rel_pos", False) else None
        if rel is not None:
            if rel.ddist.data_ptr() != ea.d_rel_bias_dist:
                raise L.CarelError("internal: the relative-position gradient buffer was re-allocated between forward and backward")
            rel.ddist[:int(ea.batch) * NH].zero_()
        lag = 1 if self.overlap_wgrad else 0     # with the side stream a layer completes one call late (include/carel_hip.h)
        for l in range(self.cfg.layers - 1, -1, -1):
            L.check(lib.carel_encoder_backward_layer(C.byref(ea), l, st), "carel_encoder_backward_layer")
            if l + lag < self.cfg.layers:
                layer_ready(l + lag)
        if lag:
            L.check(lib.carel_encoder_backward_join(C.byref(ea), st), "carel_encoder_backward_join")
            layer_ready(0)
        L.check(lib.carel_encoder_backward_embeddings(C.byref(ea), st), "carel_encoder_backward_embeddings")
        if rel is not None:      # fold the gradient by distance (every layer's attention backward added to it) into the table's buckets
            L.check(lib.carel_relpos_reduce(rel.ddist.data_ptr(), int(ea.batch), rel.bucket.data_ptr(), self._g(REL_KEY), 0, st), "carel_relpos_reduce")

    def _bind_grads(self):
        if self._grad_views is None:
            self._grad_views = {k: self._grad_view(k) for k in self._order}
        aprx = self._aprx_names
        for k, p in self._named.items():
            if k not in aprx:               # the approximation net's gradients belong to _AprxLoss.backward
                p.grad = self._grad_views[k]

    # ------------------------------------------------------------------ public API (reference surface)
    def forward(self, input_ids, att_masks, token_type_ids, emotion_labels, cause_labels, pair_labels, content_bow, iteration,
                seq_lengths=None):
        """Reference `forward` (:184-263): returns the scalar `vae_and_classifier_loss`.
        seq_lengths (optional extension): host list of attended lengths per pair; saves the device->host read that the
        padding-skipping path otherwise needs to learn them from `att_masks`."""
        c = self._make_call(input_ids, att_masks, token_type_ids, emotion_labels, cause_labels, pair_labels, content_bow, iteration,
                            training=torch.is_grad_enabled(), seq_lengths=seq_lengths)
        self._last_call = c
        vi = bool(self._aprx_names)
        D = self.opt.ec_dim
        if torch.is_grad_enabled():
            anchor = self._flat.new_zeros((), requires_grad=True)
            loss, z = _TrainLoss.apply(anchor, self, c)
            c.z_out = z
            if vi:          # drl_classifier_ec_vi.py:263: (sampled_emotion_emb, sampled_cause_emb, ec_aprx_loss, vae_and_classifier_loss)
                z_e, z_c = z[:, :D], z[:, D:]
                return z_e, z_c, self.get_ec_aprx_loss(z_e, z_c), loss
            return loss
        self._run_forward(c, training=False)
        loss = c.buf.terms[8].clone()
        if vi:
            z = c.buf.z.clone()
            return z[:, :D], z[:, D:], self.get_ec_aprx_loss(z[:, :D], z[:, D:]), loss
        return loss

    def sampled_embeddings(self):
        """(z_e, z_c) of the most recent training forward, connected to autograd: any extra loss term built on them
        back-propagates into the encoder together with the returned loss (extension; the VI variant returns them itself)."""
        z = self._last_call.z_out
        D = self.opt.ec_dim
        return z[:, :D], z[:, D:]

    def forward_terms(self, *args, **kw):
        """Added introspection entry point: the loss plus every term and the latent means (no autograd)."""
        with torch.no_grad():
            self.forward(*args, **kw)
        c = self._last_call
        t = c.buf.terms.clone()
        D = self.opt.ec_dim
        lat = c.buf.lat.clone()
        out = {n: t[i] for i, n in enumerate(self.TERM_NAMES)}
        out.update(mu_e=lat[:, :D], lv_e=lat[:, D:2 * D], mu_c=lat[:, 2 * D:3 * D], lv_c=lat[:, 3 * D:], pooled=c.buf.pooled.clone(),
                   z=c.buf.z.clone())
        return out

    def last_terms(self):
        """Terms of the most recent forward (device tensor, no sync)."""
        return {n: self._last_call.buf.terms[i] for i, n in enumerate(self.TERM_NAMES)}

    def pair_probabilities(self, input_ids, att_masks, token_type_ids, chunk=1024):
        """sigmoid(pair_classifier([z_e, z_c])) with fresh noise even in eval mode (ref :277-282, quirk Q6).  The reference feeds the
        whole test set as one batch (:958); here it goes through the encoder in chunks (inference workspace: one layer's activations
        for `chunk` pairs).  Measured on 2 048 ECPE-shaped pairs (tools/bench_infer.py): 73 / 97 / 113 / 120 k pairs/s at chunks of
        128 / 256 / 512 / 1 024 -- larger GEMM grids and fewer length read-backs."""
        self._require_cuda()
        ops._chk_cuda(input_ids, att_masks, token_type_ids)
        dev = input_ids.device
        eps_e, eps_c = self._draw_noise(dev)
        N, S = input_ids.shape
        out = torch.empty(N, device=dev, dtype=torch.float32)
        self._refresh_shadow()
        lib = L.load()
        for s in range(0, N, chunk):
            ids = input_ids[s:s + chunk]
            B = ids.shape[0]
            Bp = self._padded_batch(B, S)
            ids = self._prep_ids(ids, Bp)
            att = self._prep_ids(att_masks[s:s + chunk], Bp)
            tt = None if token_type_ids is None else self._prep_ids(token_type_ids[s:s + chunk], Bp)
            ws = self._workspace(Bp, S, inference=True)
            pack = self._pack_info(att, B, Bp, S)
            cls = self._cls_info(B, Bp, S, pack, dev)
            ea = self._encoder_args(ids, att, tt, ws, Bp, S, True, False, 0, 0, pack, cls)
            L.check(lib.carel_encoder_forward(C.byref(ea), L.current_stream()), "carel_encoder_forward")
            x_last_ptr = lib.carel_encoder_x_last(C.byref(ea))
            key = ("tail", B, S)
            buf = self._ws.get(key)
            if buf is None:
                buf = ops.TailBuffers(B, S, self.opt.ec_dim, self.opt.e_num_class, self.opt.pair_bow_dim, dev, rows=Bp * S)
                self._ws[key] = buf
            W, _ = self._tail_weights()
            ta = ops.tail_args(buf, SimpleNamespace(data_ptr=lambda: x_last_ptr), W, None, None, None, self.opt, 1.0,
                               cls_rows=cls.compact if cls is not None else (None if pack is None else pack.cu))
            ta._keep = (pack, cls)
            ops.tail_latents(ta)
            out[s:s + B] = ops.pair_probs(buf.lat, eps_e, eps_c, W["pair_classifier.weight"], W["pair_classifier.bias"], self.opt.ec_dim)
        return out

    def get_pair_preds(self, input_ids, att_masks, token_type_ids):
        """Reference `get_pair_preds` (:265-282): nested python list [[0.|1.], ...]."""
        prob = self.pair_probabilities(input_ids, att_masks, token_type_ids)
        return prob.reshape(-1, 1).cpu().detach().numpy().round().tolist()

    # reference helper kept for callers that use it directly (:515-523)
    def get_annealed_weight(self, iteration, lambda_weight):
        return (math.tanh((iteration - self.opt.kl_ann_iterations * 1.5) / (self.opt.kl_ann_iterations / 3)) + 1) * lambda_weight


class FusedAdam:
    """`torch.optim.Adam(model.get_params(), lr)` (ref :936) as ONE HIP kernel over the flat buffer.
    Same defaults and update order; also rewrites the bf16 shadow weights the GEMMs read.  The reference's
    optimiser object is only used through zero_grad()/step() (:840-842), which this class provides.

    fuse_into_backward=True (opt-in): the update of encoder layer l is enqueued on the library's auxiliary stream as
    soon as `loss.backward()` has finished that layer's gradients, so the memory-bound optimiser pass runs beside the
    remaining backward GEMMs; `step()` then only updates what is left (embeddings, pooler, heads) and joins.  Same
    arithmetic, same results -- but the weights move during backward(), so it requires the reference's call pattern
    zero_grad() -> backward() -> step() (no gradient accumulation, no use of the gradients to decide whether to step).
    Under DataParallel (RCCL) each layer's update additionally waits for that layer's gradient all-reduce."""

    def __init__(self, model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, fuse_into_backward=False, param_range=None, params=None):
        """param_range / params (both or neither): a contiguous [lo, hi) slice of the flat buffer and the Parameters in
        it, for models whose get_params() returns several optimiser groups (drl_classifier_en.py:357-376)."""
        model._require_cuda()
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.step_count = 0
        self._lo, self._hi = (0, model._n_opt) if param_range is None else param_range
        self._clear_all = params is None      # the single-optimiser scripts: zero_grad() clears every .grad of the model
        if params is None:
            params = model.get_params() if not model._aprx_names else model.get_params()[1]
        self._params = list(params)
        self.exp_avg = torch.zeros(self._hi - self._lo, device=model._flat.device, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.grad_scale = 1.0
        self.param_groups = [dict(params=self._params, lr=lr, betas=betas, eps=eps)]
        self._done = []                  # [lo, hi) ranges already updated for the coming step()
        self._skip_count = None
        self._aux = None
        if fuse_into_backward:
            # Two library streams.  Without collectives the updates are queued on the WEIGHT-GRADIENT stream (0), behind the layer's grouped
            # weight-gradient launch that already runs there: measured on one box, alternating (tools/ab_adam_stream.sh, round 4): dense step
            # 7.80-7.85 ms with the update in step(), 7.69-7.72 on the auxiliary stream (1), 7.63-7.71 on the weight-gradient stream;
            # ECPE-shaped 4.38-4.40 / 4.34 / 4.27-4.31.  Under DataParallel an update waits for its bucket's all-reduce, which must not
            # hold up the weight gradients queued behind it: those updates go to the auxiliary stream.
            ptrs = [L.load().carel_side_stream(i) for i in (0, 1)]
            if not all(ptrs):
                raise L.CarelError("carel_side_stream: " + L.load().carel_last_error().decode("utf8", "replace"))
            self._side = torch.cuda.ExternalStream(ptrs[0], device=model._flat.device)
            self._aux = torch.cuda.ExternalStream(ptrs[1], device=model._flat.device)
            self._used = set()
            self._ev = torch.cuda.Event()
            offs, nl = model._offs, model.cfg.layers
            starts = [offs[f"encoder.encoder.layer.{l}.attention.self.query.weight"] for l in range(nl)]
            starts.append(offs["encoder.pooler.dense.weight"])
            self._layer_ranges = [(starts[l], starts[l + 1]) for l in range(nl)]
            model._adam_hook = self

    def zero_grad(self, set_to_none=True):
        for p in (self.model._named.values() if self._clear_all else self._params):
            p.grad = None

    def _launch(self, lo, hi, stream, with_skip):
        m = self.model
        a = L.AdamArgs()
        a.param, a.grad = m._flat.data_ptr() + 4 * lo, m._flat_grad.data_ptr() + 4 * lo
        a.exp_avg, a.exp_avg_sq = self.exp_avg.data_ptr() + 4 * (lo - self._lo), self.exp_avg_sq.data_ptr() + 4 * (lo - self._lo)
        a.shadow_bf16 = m._shadow.data_ptr() + 2 * lo
        a.n, a.step = hi - lo, self.step_count + 1
        a.lr, a.beta1, a.beta2, a.eps = self.param_groups[0]["lr"], self.betas[0], self.betas[1], self.eps
        a.grad_scale = self.grad_scale
        a.skip_lo, a.skip_hi, a.skip_flag = 0, 0, None
        call = getattr(m, "_last_call", None)
        if with_skip and m._has_pair_skip and call is not None and lo <= m._pair_lo and m._pair_hi <= hi:
            # the pair head keeps its weights/moments when its loss term was replaced by 0
            a.skip_lo, a.skip_hi = m._pair_lo - lo, m._pair_hi - lo
            off = L.load().carel_tail_pair_dead_offset(call.B, m.opt.ec_dim, m.opt.pair_bow_dim)
            a.skip_flag = call.buf.work.data_ptr() + off * 4
            if self._skip_count is None:       # frozen steps of the pair head so far (torch: that parameter's own step counter lags)
                self._skip_count = torch.zeros(1, device=m._flat.device, dtype=torch.float32)
            a.skip_count = self._skip_count.data_ptr()
        L.check(L.load().carel_adam_step(C.byref(a), stream), "carel_adam_step")

    def _layer_ready(self, layer, after=None):
        """Called by the model's backward when encoder layer `layer` has all its gradients on the main stream; under
        DataParallel `after` is the work handle of the layer's gradient all-reduce (RCCL), which the update waits for."""
        self._range_ready(self._layer_ranges[layer], after)

    def _range_ready(self, rng, after=None):
        """Update the flat range [lo, hi) on the auxiliary stream, after the main stream's work so far (`after` None) or after the
        collective behind the handle `after` (its wait() orders the auxiliary stream only)."""
        lo, hi = rng
        if after is not None:
            st = self._aux
            with torch.cuda.stream(st):
                after.wait()                 # stream-side wait: the auxiliary stream blocks until the collective is done
        else:
            st = self._aux if os.environ.get("CAREL_ADAM_STREAM") == "1" else self._side      # (the variable: A/B tool only, tools/ab_adam_stream.sh)
            self._ev.record()
            st.wait_event(self._ev)
        self._used.add(st)
        self._launch(lo, hi, C.c_void_p(st.cuda_stream), with_skip=False)
        self._done.append((lo, hi))

    def _join(self):
        if self._aux is not None and self._done:
            for st in self._used:
                torch.cuda.current_stream().wait_stream(st)
            self._used.clear()

    def step(self):
        m = self.model
        lo = self._lo
        for dlo, dhi in sorted(self._done):             # whatever backward() has not updated yet
            if dlo > lo:
                self._launch(lo, dlo, L.current_stream(), with_skip=True)
            lo = max(lo, dhi)
        if lo < self._hi:
            self._launch(lo, self._hi, L.current_stream(), with_skip=True)
        self._join()
        self._done = []
        self.step_count += 1
        m.mark_shadow_fresh()

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, lr=self.lr,
                    pair_head_frozen_steps=None if self._skip_count is None else self._skip_count.clone())

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        fr = sd.get("pair_head_frozen_steps")
        self._skip_count = None if fr is None else fr.to(self.exp_avg.device, torch.float32).reshape(1).clone()
