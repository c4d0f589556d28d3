"""CPU oracle for the CAREL-VAE training hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The shipped path (`carel_vae_amd`) never does.

It is a from-scratch, pure-PyTorch-CPU fp32 restatement of the reference's algorithm for the path
`drl_classifier_ec_mmd_final_mul.py` :184-263 (forward), :841-842 (backward + Adam), written from the
formulas, not copied.  The encoder arithmetic lives in the third-party `transformers` package
(un-pinned by the reference; 5.15.0 installed here): `BertModel` / `RobertaModel`, eager attention.
Its published recipe is restated in `encoder_forward` below.

Parity pin: `tests/golden/gen_golden.py` executes the reference's own `DrlClassifier`, `MMDStatistic`
and `pdist` class bodies (AST-extracted from /root/reference at generation time, never copied) against
`transformers.BertModel(BertConfig(...))` with seeded weights and records inputs/outputs in
`tests/golden/*.npz`; `tests/test_oracle_golden.py` holds this oracle to those vectors.  The reference
itself ships no tests or known-answer vectors for this path.

Reference line numbers below are into /root/reference/drl_classifier_ec_mmd_final_mul.py unless a
file name is given.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import numpy as np
import torch

# --------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------


@dataclass
class EncoderConfig:
    """BERT-base geometry (`hfl/chinese-roberta-wwm-ext` :159 == bert-base-chinese architecture)."""
    vocab_size: int = 21128
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    ln_eps: float = 1e-12
    variant: str = "bert"          # "bert" (:159), "roberta" (:162) or "mpnet" (en_ec_sentence_transformer.py:22: all-mpnet-base-v2)
    pad_id: int = 0                # roberta: 1
    hidden_dropout: float = 0.1    # HF defaults, active under model.train() (:822)
    attn_dropout: float = 0.1
    rel_pos: bool = False          # MPNet: attention_scores += relative_attention_bias[bucket(key - query)] (shared by all layers)

    @staticmethod
    def mpnet_base() -> "EncoderConfig":
        """transformers MPNetConfig defaults (microsoft/mpnet-base, all-mpnet-base-v2): no token types, RoBERTa position ids"""
        return EncoderConfig(vocab_size=30527, max_pos=514, type_vocab=1, ln_eps=1e-5, variant="mpnet", pad_id=1, rel_pos=True)

    @staticmethod
    def roberta_base() -> "EncoderConfig":
        return EncoderConfig(vocab_size=50265, max_pos=514, type_vocab=1, ln_eps=1e-5,
                             variant="roberta", pad_id=1)


@dataclass
class Opt:
    """The argparse namespace of the reference (:30-61), defaults identical."""
    language: str = "zh"
    max_len: int = 128
    e_num_class: int = 6
    c_num_class: int = 1
    pair_num_class: int = 1
    ec_dim: int = 24
    bert_dim: int = 768
    kl_ann_iterations: int = 20000
    epochs: int = 20
    batch_size: int = 64
    ec_kl_lambda: float = 0.03
    label_smoothing: float = 0.1
    mmd_loss_weight: float = 30.0
    emo_mul_loss_weight: float = 10.0
    cau_mul_loss_weight: float = 10.0
    pair_mul_loss_weight: float = 30.0
    dropout: float = 0.5
    epsilon: float = 1e-8
    vae_lr: float = 1e-5
    aprx_lr: float = 0.003          # drl_classifier_ec_vi.py:51 (VI ablation only)
    pair_bow_dim: int = 23771
    self_iteration: int = 50
    self_epochs: int = 10
    self_strategy: str = "random"
    best_model_path: str = "ECPE_model/best_cause_pair_model"
    model_id: str = "oracle"


# --------------------------------------------------------------------------------------------
# counter-based dropout masks shared bit-for-bit with the HIP kernels (csrc/carel_rng.h)
# --------------------------------------------------------------------------------------------

SITE_EMBED = 0
SITE_TAIL_EMO, SITE_TAIL_CAU, SITE_TAIL_PAIR = 100, 101, 102


def site_attn_probs(layer: int) -> int:
    return 1 + 3 * layer


def site_attn_out(layer: int) -> int:
    return 2 + 3 * layer


def site_ffn_out(layer: int) -> int:
    return 3 + 3 * layer


def _mix32(x: np.ndarray) -> np.ndarray:
    """lowbias32 integer hash on uint32 lanes (wraps mod 2^32)."""
    x = x.astype(np.uint64)
    m = np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & m
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & m
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def dropout_threshold(p: float) -> int:
    """keep element iff hash >= threshold; threshold = floor(p * 2^32)."""
    return int(min(max(p, 0.0), 1.0) * 4294967296.0) & 0xFFFFFFFF if p < 1.0 else 0xFFFFFFFF


def dropout_keep(seed: int, site: int, idx: np.ndarray, p: float) -> np.ndarray:
    """Boolean keep-mask for linear element indices `idx` (uint32) at dropout `site`."""
    key = _mix32(np.array([(seed + site * 0x9E3779B9) & 0xFFFFFFFF], dtype=np.uint32))[0]
    # two consecutive elements share one hash: element j takes the 16-bit half (j & 1) of mix32((j >> 1) ^ key) against the upper
    # 16 bits of the threshold (csrc/carel_common.h dropout_mult)
    j = idx.astype(np.uint32)
    h = _mix32((j >> np.uint32(1)) ^ key)
    piece = np.where((j & np.uint32(1)) != 0, h >> np.uint32(16), h & np.uint32(0xFFFF))
    return piece >= np.uint32(dropout_threshold(p) >> 16)


def dropout_scale_mask(seed: Optional[int], site: int, shape, p: float, row_offset: int = 0,
                       row_elems: Optional[int] = None) -> Optional[torch.Tensor]:
    """Float mask (0 or 1/(1-p)) of `shape`; `row_offset` shifts the leading (sample) index so that a
    data-parallel shard reproduces the masks of the unsharded batch."""
    if seed is None or p <= 0.0:
        return None
    n = int(np.prod(shape))
    if row_elems is None:
        row_elems = n // shape[0]
    idx = (np.arange(n, dtype=np.uint64) + np.uint64(row_offset * row_elems)).astype(np.uint32)
    keep = dropout_keep(seed, site, idx, p).reshape(shape)
    return torch.from_numpy(keep.astype(np.float32) * np.float32(1.0 / (1.0 - p)))


# --------------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------------


def param_shapes(cfg: EncoderConfig, opt: Opt) -> Dict[str, tuple]:
    """state_dict key -> shape, in `nn.Module` registration order of the reference (:159-179):
    encoder (HF BertModel order), 4 latent heads, 3 classifiers, decoder."""
    H, I = cfg.hidden, cfg.intermediate
    s: Dict[str, tuple] = {}
    e = "encoder.embeddings."
    s[e + "word_embeddings.weight"] = (cfg.vocab_size, H)
    s[e + "position_embeddings.weight"] = (cfg.max_pos, H)
    s[e + "token_type_embeddings.weight"] = (cfg.type_vocab, H)
    s[e + "LayerNorm.weight"] = (H,)
    s[e + "LayerNorm.bias"] = (H,)
    if cfg.rel_pos:
        s["encoder.encoder.relative_attention_bias.weight"] = (32, cfg.heads)
    for l in range(cfg.layers):
        p = f"encoder.encoder.layer.{l}."
        for n in ("query", "key", "value"):
            s[p + f"attention.self.{n}.weight"] = (H, H)
            s[p + f"attention.self.{n}.bias"] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    s["encoder.pooler.dense.weight"] = (H, H)
    s["encoder.pooler.dense.bias"] = (H,)
    D = opt.ec_dim
    for n in ("emotion_mu", "emotion_log_var", "cause_mu", "cause_log_var"):
        s[n + ".weight"] = (D, opt.bert_dim)
        s[n + ".bias"] = (D,)
    s["emotion_classifier.weight"] = (opt.e_num_class, D)
    s["emotion_classifier.bias"] = (opt.e_num_class,)
    s["cause_classifier.weight"] = (opt.c_num_class, D)
    s["cause_classifier.bias"] = (opt.c_num_class,)
    s["pair_classifier.weight"] = (opt.pair_num_class, 2 * D)
    s["pair_classifier.bias"] = (opt.pair_num_class,)
    s["decoder.weight"] = (opt.pair_bow_dim, 2 * D)
    s["decoder.bias"] = (opt.pair_bow_dim,)
    return s


UNOPTIMISED_PREFIXES = ("emotion_mu.", "emotion_log_var.", "cause_mu.", "cause_log_var.")


def optimised_keys(cfg: EncoderConfig, opt: Opt):
    """Keys in `get_params()` order (:292-295): encoder, decoder, emotion/cause/pair classifiers.
    The four latent heads are absent."""
    keys = list(param_shapes(cfg, opt).keys())
    enc = [k for k in keys if k.startswith("encoder.")]
    rest = []
    for pre in ("decoder.", "emotion_classifier.", "cause_classifier.", "pair_classifier."):
        rest += [k for k in keys if k.startswith(pre)]
    return enc + rest


def init_params(cfg: EncoderConfig, opt: Opt, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights (pretrained checkpoints are unavailable offline).
    numpy `RandomState` (frozen MT19937 stream) so fixtures regenerate identically anywhere.
    Encoder: normal(0, 0.02), LayerNorm 1/0, biases small-random (HF inits biases to 0; non-zero
    biases make bias handling observable in parity tests).  Heads: uniform(+-1/sqrt(fan_in)) like
    `nn.Linear`."""
    rs = np.random.RandomState(seed)
    out: Dict[str, torch.Tensor] = {}
    for k, shp in param_shapes(cfg, opt).items():
        if k.startswith("encoder."):
            if "LayerNorm.weight" in k:
                a = 1.0 + 0.05 * rs.standard_normal(shp)
            elif "LayerNorm.bias" in k or k.endswith(".bias"):
                a = 0.02 * rs.standard_normal(shp)
            elif "relative_attention_bias" in k:
                a = 0.5 * rs.standard_normal(shp)          # large enough to move attention visibly in the parity tests
            elif cfg.variant == "mpnet" and "token_type_embeddings" in k:
                a = np.zeros(shp)                          # MPNet has no token types: the row exists only as a zero placeholder
            else:
                a = 0.02 * rs.standard_normal(shp)
        else:
            fan_in = shp[1] if len(shp) == 2 else param_shapes(cfg, opt)[k.replace(".bias", ".weight")][1]
            bound = 1.0 / math.sqrt(fan_in)
            a = rs.uniform(-bound, bound, size=shp)
        out[k] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return out


# --------------------------------------------------------------------------------------------
# encoder (restatement of transformers BertModel / RobertaModel forward, eager attention)
# --------------------------------------------------------------------------------------------

Quant = Optional[Callable[[torch.Tensor], torch.Tensor]]


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bfloat16 and back: emulates the HIP path's bf16 storage points."""
    return x.to(torch.bfloat16).to(torch.float32)


def _q(x, quant: Quant):
    return x if quant is None else quant(x)


class _RoundSTE(torch.autograd.Function):
    """bf16 rounding in the forward pass, identity in the backward pass (plain `bf16_round` also rounds the GRADIENT that flows back
    through its bf16 intermediate -- at a point where the HIP path does not round)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _GradRound(torch.autograd.Function):
    """identity in the forward pass; the gradient is rounded to bf16 in the backward pass"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _GeluSaved(torch.autograd.Function):
    """gelu(u) and its derivative as the FFN1 epilogue stores them: both evaluated on the bf16-rounded pre-activation, both rounded to
    bf16; the backward pass multiplies by the SAVED derivative (csrc/gemm_epilogue.h, CAREL_EPI_BIAS_GELU_DG / CAREL_EPI_MUL_BF16)."""

    @staticmethod
    def forward(ctx, u):
        ur = u.to(torch.bfloat16).to(torch.float32)
        cdf = 0.5 * (1.0 + torch.erf(ur * (1.0 / math.sqrt(2.0))))
        dg = cdf + ur * torch.exp(-0.5 * ur * ur) * (1.0 / math.sqrt(2.0 * math.pi))
        ctx.save_for_backward(dg.to(torch.bfloat16).to(torch.float32))
        return (ur * cdf).to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (dg,) = ctx.saved_tensors
        return g * dg


class Bf16Hip:
    """`quant=` object that emulates the bf16 storage points of the HIP path in BOTH directions (tests of the backward pass): operands
    rounded in the forward pass with a straight-through gradient; the gradient signals rounded where the library stores them in bf16 --
    d(out-projection / FFN2 output) after the dropout mask (dyb, dyb2), d(FFN1 pre-activation) (du), d(context) (dctx), dS and
    d(q, k, v) (dqkv) -- and GELU with its saved, bf16-rounded derivative (csrc/encoder.hip, carel_encoder_backward_layer)."""

    def __call__(self, x):
        return _RoundSTE.apply(x)

    @staticmethod
    def grad(x):
        return _GradRound.apply(x)

    @staticmethod
    def gelu(u):
        return _GeluSaved.apply(u)


bf16_hip = Bf16Hip()


def layer_norm(x, w, b, eps):
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x, w, b, quant: Quant = None):
    """y = x W^T + b.  Under `quant` the two GEMM operands are rounded (bf16 MFMA inputs) while the
    accumulation and the bias add stay fp32."""
    return _q(x, quant) @ _q(w, quant).t() + b


def position_ids(ids: torch.Tensor, cfg: EncoderConfig) -> torch.Tensor:
    B, S = ids.shape
    if cfg.variant in ("roberta", "mpnet"):   # transformers modeling_roberta.py / modeling_mpnet.py create_position_ids_from_input_ids
        m = (ids != cfg.pad_id).to(torch.int64)
        return torch.cumsum(m, dim=1) * m + cfg.pad_id
    return torch.arange(S, dtype=torch.int64).unsqueeze(0).expand(B, S)


def mpnet_relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """transformers MPNetEncoder.relative_position_bucket, expression by expression (float32 log, truncation)."""
    ret = 0
    n = -relative_position
    num_buckets //= 2
    ret += (n < 0).to(torch.long) * num_buckets
    n = torch.abs(n)
    max_exact = num_buckets // 2
    is_small = n < max_exact
    val_if_large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).to(torch.long)
    val_if_large = torch.min(val_if_large, torch.full_like(val_if_large, num_buckets - 1))
    ret += torch.where(is_small, n, val_if_large)
    return ret


def mpnet_position_bias(table: torch.Tensor, S: int) -> torch.Tensor:
    """MPNetEncoder.compute_position_bias: [1, heads, S, S], entry [h, i, j] = table[bucket(j - i), h]"""
    ctx = torch.arange(S, dtype=torch.long)[:, None]
    mem = torch.arange(S, dtype=torch.long)[None, :]
    rp = mpnet_relative_position_bucket(mem - ctx)
    return table[rp].permute(2, 0, 1).unsqueeze(0)


def encoder_forward(P: Dict[str, torch.Tensor], ids, att_mask, token_type, cfg: EncoderConfig,
                    train: bool = False, seed: Optional[int] = None, row_offset: int = 0,
                    quant: Quant = None, taps: Optional[dict] = None) -> torch.Tensor:
    """pooler_output [B,H] (the only encoder output the hot path consumes, :202-206).

    `train`+`seed` switch on the three BERT dropouts (embeddings, attention probabilities, the two
    sub-layer outputs) with the counter-based masks of `dropout_keep`.  `quant`, when given, is applied
    exactly where the HIP path stores bf16 (GEMM operands / saved activations); None = pure fp32.
    """
    B, S = ids.shape
    H, nh = cfg.hidden, cfg.heads
    dh = H // nh
    e = "encoder.embeddings."
    x = P[e + "word_embeddings.weight"][ids] + P[e + "position_embeddings.weight"][position_ids(ids, cfg)]
    if cfg.variant != "mpnet":     # MPNetEmbeddings has no token types (the key exists here as an all-zero, never-updated row)
        x = x + P[e + "token_type_embeddings.weight"][token_type]
    x = layer_norm(x, P[e + "LayerNorm.weight"], P[e + "LayerNorm.bias"], cfg.ln_eps)
    ph, pa = (cfg.hidden_dropout, cfg.attn_dropout) if train else (0.0, 0.0)
    m = dropout_scale_mask(seed, SITE_EMBED, (B, S, H), ph, row_offset)
    if m is not None:
        x = x * m
    gq = getattr(quant, "grad", None) or (lambda t: t)          # Bf16Hip: gradient signals rounded where the library stores them in bf16
    neg = torch.finfo(torch.float32).min
    mask_add = (1.0 - att_mask.to(torch.float32))[:, None, None, :] * neg      # [B,1,1,S]
    pos_bias = mpnet_position_bias(P["encoder.encoder.relative_attention_bias.weight"], S) if cfg.rel_pos else None
    if taps is not None:
        taps["x0"] = x
    for l in range(cfg.layers):
        p = f"encoder.encoder.layer.{l}."
        q = linear(x, P[p + "attention.self.query.weight"], P[p + "attention.self.query.bias"], quant)
        k = linear(x, P[p + "attention.self.key.weight"], P[p + "attention.self.key.bias"], quant)
        v = linear(x, P[p + "attention.self.value.weight"], P[p + "attention.self.value.bias"], quant)
        q, k, v = (_q(gq(t), quant).view(B, S, nh, dh).transpose(1, 2) for t in (q, k, v))
        s = gq(q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
        if pos_bias is not None:   # MPNetSelfAttention: scores / sqrt(d), += position_bias, += attention_mask
            s = s + pos_bias
        s = s + mask_add
        pr = torch.softmax(s, dim=-1)
        m = dropout_scale_mask(seed, site_attn_probs(l), (B, nh, S, S), pa, row_offset)
        if m is not None:
            pr = pr * m
        ctx = (_q(pr, quant) @ v).transpose(1, 2).reshape(B, S, H)
        ctx = _q(gq(ctx), quant)
        a = gq(linear(ctx, P[p + "attention.output.dense.weight"], P[p + "attention.output.dense.bias"], quant))
        m = dropout_scale_mask(seed, site_attn_out(l), (B, S, H), ph, row_offset)
        if m is not None:
            a = a * m
        x1 = layer_norm(a + x, P[p + "attention.output.LayerNorm.weight"],
                        P[p + "attention.output.LayerNorm.bias"], cfg.ln_eps)
        u = gq(linear(x1, P[p + "intermediate.dense.weight"], P[p + "intermediate.dense.bias"], quant))
        g = quant.gelu(u) if hasattr(quant, "gelu") else _q(gelu_erf(u), quant)
        f = gq(linear(g, P[p + "output.dense.weight"], P[p + "output.dense.bias"], quant))
        m = dropout_scale_mask(seed, site_ffn_out(l), (B, S, H), ph, row_offset)
        if m is not None:
            f = f * m
        x = layer_norm(f + x1, P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], cfg.ln_eps)
        if taps is not None:
            taps[f"x{l + 1}"] = x
    cls = x[:, 0, :]
    pooled = torch.tanh(cls @ P["encoder.pooler.dense.weight"].t() + P["encoder.pooler.dense.bias"])
    return pooled


# --------------------------------------------------------------------------------------------
# statistic heads: RBF-MMD (:537-596), HSIC (drl_classifier_ec_hsic.py:529-547)
# --------------------------------------------------------------------------------------------


def pdist(sample_1, sample_2, norm=2, eps=1e-5):
    """Pairwise L2 distances with the reference's eps-inside-sqrt and abs() (:580-589)."""
    n1 = (sample_1 * sample_1).sum(dim=1, keepdim=True)
    n2 = (sample_2 * sample_2).sum(dim=1, keepdim=True)
    d2 = n1 + n2.t() - 2.0 * (sample_1 @ sample_2.t())
    return torch.sqrt(eps + d2.abs())


def mmd_statistic(s1, s2, alphas, ret_matrix=False):
    """Unbiased-within / biased-cross MMD^2 estimate of `MMDStatistic.__call__` (:547-569)."""
    n1, n2 = s1.shape[0], s2.shape[0]
    a00 = 1.0 / (n1 * (n1 - 1))
    a11 = 1.0 / (n2 * (n2 - 1))
    a01 = -1.0 / (n1 * n2)
    z = torch.cat((s1, s2), dim=0)
    d = pdist(z, z)
    kern = None
    for a in alphas:
        ka = torch.exp(-a * d ** 2)
        kern = ka if kern is None else kern + ka
    k1, k2, k12 = kern[:n1, :n1], kern[n1:, n1:], kern[:n1, n1:]
    mmd = 2 * a01 * k12.sum() + a00 * (k1.sum() - torch.trace(k1)) + a11 * (k2.sum() - torch.trace(k2))
    return (mmd, kern) if ret_matrix else mmd


def hsic_statistic(x, y, s_x=1.0, s_y=1.0):
    """`HSIC` of drl_classifier_ec_hsic.py:540-547: tr(L H K H)/(m-1)^2 with Gaussian kernels on
    squared distances (no eps, no abs; :529-537).  The reference builds H in float64 but casts every
    operand back with `.float()` before the matmuls (:546), so the arithmetic is fp32."""
    def pw(t):
        inst = (t * t).sum(dim=-1).reshape(-1, 1)
        return -2 * (t @ t.t()) + inst + inst.t()
    m = x.shape[0]
    K = torch.exp(-pw(x) / s_x)
    L = torch.exp(-pw(y) / s_y)
    Hm = torch.eye(m) - (1.0 / m) * torch.ones((m, m))
    return torch.trace(L @ (Hm @ (K @ Hm))) / ((m - 1) ** 2)


# --------------------------------------------------------------------------------------------
# VAE tail + losses (:209-261)
# --------------------------------------------------------------------------------------------


def kl_anneal_weight(iteration: int, opt: Opt) -> float:
    """:515-523, host double arithmetic."""
    return (math.tanh((iteration - opt.kl_ann_iterations * 1.5) / (opt.kl_ann_iterations / 3)) + 1) * opt.ec_kl_lambda


def bce_prob(p, t):
    """nn.BCELoss elementwise: logs clamped at -100."""
    return -(t * torch.clamp(torch.log(p), min=-100.0) + (1.0 - t) * torch.clamp(torch.log(1.0 - p), min=-100.0))


def bce_logits_posw(x, t, pw):
    """nn.BCEWithLogitsLoss(pos_weight) elementwise."""
    lw = (pw - 1.0) * t + 1.0
    return (1.0 - t) * x + lw * (torch.log1p(torch.exp(-x.abs())) + torch.clamp(-x, min=0.0))


def tail_forward(P, pooled, emo_labels, cau_labels, pair_labels, bow, iteration: int, opt: Opt,
                 eps_e: torch.Tensor, eps_c: torch.Tensor, train: bool = False, seed: Optional[int] = None,
                 row_offset: int = 0, global_label_sum: Optional[float] = None,
                 global_n: Optional[int] = None, disentangle: str = "mmd", emotion_head: str = "ce") -> Dict[str, torch.Tensor]:
    """Everything after `pooler_output` (:209-261).  Returns every term separately plus `loss`.

    eps_e / eps_c: the two `[ec_dim]` noise vectors of `sample_prior` (:350; one vector shared by the
    whole batch, emotion drawn first).  Dropout p=opt.dropout on the three classifier inputs when `train`.
    """
    B = pooled.shape[0]
    mu_e = pooled @ P["emotion_mu.weight"].t() + P["emotion_mu.bias"]
    lv_e = pooled @ P["emotion_log_var.weight"].t() + P["emotion_log_var.bias"]
    mu_c = pooled @ P["cause_mu.weight"].t() + P["cause_mu.bias"]
    lv_c = pooled @ P["cause_log_var.weight"].t() + P["cause_log_var.bias"]
    z_e = mu_e + eps_e * torch.exp(lv_e)      # :351 -- std = exp(log_var) (quirk Q2)
    z_c = mu_c + eps_c * torch.exp(lv_c)
    z = torch.cat((z_e, z_c), dim=1)          # generative_emb == pair_emb (:219-220)
    pd = opt.dropout if train else 0.0
    D = opt.ec_dim

    def drop(t, site, width):
        m = dropout_scale_mask(seed, site, (B, width), pd, row_offset)
        return t if m is None else t * m

    # emotion head :461-476
    logit_e = drop(z_e, SITE_TAIL_EMO, D) @ P["emotion_classifier.weight"].t() + P["emotion_classifier.bias"]
    ls = opt.label_smoothing
    if emotion_head == "bce":          # 1-logit head of drl_classifier_ec_hsic.py:455-470 / ec_vi
        pe = torch.sigmoid(logit_e)
        emo = bce_prob(pe, emo_labels.view(-1, 1).to(torch.float32) * (1 - ls) + ls / 1).mean()
    else:
        lse = torch.logsumexp(logit_e, dim=1)
        emo = (lse - logit_e.gather(1, emo_labels.view(-1, 1)).squeeze(1)).mean()
    # cause head :478-492
    ls = opt.label_smoothing
    pc = torch.sigmoid(drop(z_c, SITE_TAIL_CAU, D) @ P["cause_classifier.weight"].t() + P["cause_classifier.bias"])
    cau = bce_prob(pc, cau_labels * (1 - ls) + ls / opt.c_num_class).mean()
    # disentanglement :231-233
    if disentangle == "mmd":
        mmd = mmd_statistic(z_e, z_c, [0.1])
        dis = -mmd
    elif disentangle == "hsic":
        mmd = hsic_statistic(z_e, z_c)
        dis = mmd
    else:
        mmd = torch.zeros(())
        dis = mmd
    # pair head :494-513
    xp = drop(z, SITE_TAIL_PAIR, 2 * D) @ P["pair_classifier.weight"].t() + P["pair_classifier.bias"]
    n = float(global_n if global_n is not None else B)
    sy = pair_labels.sum() if global_label_sum is None else torch.tensor(float(global_label_sum))
    pw = (n - sy) / sy
    pair = bce_logits_posw(xp, pair_labels * (1 - ls) + ls / opt.pair_num_class, pw).mean()
    if bool(torch.isinf(pair).any()):
        pair = torch.zeros(())
    # KL :525-534, :240-250
    def kl(mu, lv):
        return (-0.5 * (1 + lv - lv.exp() - mu.pow(2)).sum(dim=1)).mean()
    kl_e, kl_c = kl(mu_e, lv_e), kl(mu_c, lv_c)
    if iteration < opt.kl_ann_iterations:
        w = kl_anneal_weight(iteration, opt)
        kl_e, kl_c = w * kl_e, w * kl_c
    # reconstruction :253-254, :381-387
    prob = torch.softmax(z @ P["decoder.weight"].t() + P["decoder.bias"], dim=1)
    rec = bce_prob(prob, bow * (1 - ls) + ls / opt.pair_bow_dim).mean()
    w_dis = opt.mmd_loss_weight if disentangle == "mmd" else 1.0
    loss = w_dis * dis + opt.emo_mul_loss_weight * emo + opt.cau_mul_loss_weight * cau \
        + opt.pair_mul_loss_weight * pair + kl_e + kl_c + rec
    return dict(loss=loss, mmd=mmd, emo=emo, cau=cau, pair=pair, kl_e=kl_e, kl_c=kl_c, rec=rec,
                mu_e=mu_e, mu_c=mu_c, lv_e=lv_e, lv_c=lv_c, z_e=z_e, z_c=z_c, pair_logit=xp)


def forward_terms(P, batch: Dict[str, torch.Tensor], iteration: int, cfg: EncoderConfig, opt: Opt,
                  eps_e, eps_c, train=False, seed=None, row_offset=0, quant: Quant = None, **kw):
    """`DrlClassifier.forward` (:184-263) with every term exposed."""
    pooled = encoder_forward(P, batch["input_ids"], batch["attention_masks"], batch["token_type_ids"], cfg,
                             train=train, seed=seed, row_offset=row_offset, quant=quant)
    out = tail_forward(P, pooled, batch["emo_labels"], batch["cau_labels"], batch["labels"], batch["bow_reps"],
                       iteration, opt, eps_e, eps_c, train=train, seed=seed, row_offset=row_offset, **kw)
    out["pooled"] = pooled
    return out


def pair_preds(P, ids, att, tt, cfg, opt, eps_e, eps_c, quant: Quant = None):
    """`get_pair_preds` (:265-282): eval-mode encoder, fresh noise (quirk Q6), round(sigmoid)."""
    pooled = encoder_forward(P, ids, att, tt, cfg, quant=quant)
    mu_e = pooled @ P["emotion_mu.weight"].t() + P["emotion_mu.bias"]
    lv_e = pooled @ P["emotion_log_var.weight"].t() + P["emotion_log_var.bias"]
    mu_c = pooled @ P["cause_mu.weight"].t() + P["cause_mu.bias"]
    lv_c = pooled @ P["cause_log_var.weight"].t() + P["cause_log_var.bias"]
    z = torch.cat((mu_e + eps_e * lv_e.exp(), mu_c + eps_c * lv_c.exp()), dim=1)
    prob = torch.sigmoid(z @ P["pair_classifier.weight"].t() + P["pair_classifier.bias"])
    return prob


# --------------------------------------------------------------------------------------------
# VI / CLUB head of the ablation script drl_classifier_ec_vi.py (approximation network p(e|c))
# --------------------------------------------------------------------------------------------

VI_KEYS = tuple(f"{n}.{i}.{t}" for n in ("ec_mu", "ec_log_var") for i in (0, 2) for t in ("weight", "bias"))


def init_vi_params(opt: Opt, seed: int = 0) -> Dict[str, torch.Tensor]:
    """nn.Linear-style uniform(+-1/sqrt(fan_in)) weights for ec_mu / ec_log_var (ec_vi :156-163), RandomState stream."""
    rs = np.random.RandomState(seed)
    D = opt.ec_dim
    bound = 1.0 / math.sqrt(D)
    return {k: torch.from_numpy(rs.uniform(-bound, bound, size=(D, D) if k.endswith("weight") else (D,)).astype(np.float32))
            for k in VI_KEYS}


def vi_net(P, c):
    """get_ec_emb (ec_vi :343-348): mu = Linear-ReLU-Linear, log_var = tanh(Linear-ReLU-Linear)."""
    def mlp(n):
        h = torch.relu(c @ P[n + ".0.weight"].t() + P[n + ".0.bias"])
        return h @ P[n + ".2.weight"].t() + P[n + ".2.bias"]
    return mlp("ec_mu"), torch.tanh(mlp("ec_log_var"))


def vi_aprx_loss(P, z_e, z_c):
    """get_ec_aprx_loss (ec_vi :422-427); the cause embedding is detached."""
    mu, lv = vi_net(P, z_c.detach())
    return -((-(mu - z_e) ** 2 / lv.exp() - lv).sum(dim=1).mean(dim=0))


def vi_upper_loss(P, z_e, z_c, perm):
    """get_ec_upper_loss (ec_vi :429-440) with random_index = perm."""
    mu, lv = vi_net(P, z_c)
    positive = -(mu - z_e) ** 2 / lv.exp()
    negative = -(mu - z_e[perm.long()]) ** 2 / lv.exp()
    return (positive.sum(dim=-1) - negative.sum(dim=-1)).mean() / 2.


def vi_train_step(P, batch, iteration, epoch, cfg, opt, eps_e, eps_c, perm, st_vae: "AdamState", st_aprx: "AdamState",
                  quant: Quant = None, train=False, seed=None, emotion_head: str = "bce"):
    """The two-phase step of ec_vi :754-774: Adam(aprx_lr) on the approximation net with the aprx loss, then
    vae loss + beta * CLUB bound (with the UPDATED net) -> Adam(vae_lr) on get_params()[1].
    emotion_head: "bce" = drl_classifier_ec_vi.py (one-logit head), "ce" = drl_classifier_ec_vi_final.py (:465-477,
    six-way cross entropy, the head of the main script).  Returns (P, dict(aprx=, vae=, upper=, total=))."""
    leaf = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    out = forward_terms(leaf, batch, iteration, cfg, opt, eps_e, eps_c, train=train, seed=seed, quant=quant,
                        disentangle="none", emotion_head=emotion_head)
    aprx = vi_aprx_loss(leaf, out["z_e"], out["z_c"])
    g_aprx = torch.autograd.grad(aprx, [leaf[k] for k in VI_KEYS], retain_graph=True)
    P = dict(P)
    P = adam_step(P, dict(zip(VI_KEYS, g_aprx)), list(VI_KEYS), st_aprx, lr=opt.aprx_lr)
    net = {k: P[k] for k in VI_KEYS}
    upper = vi_upper_loss(net, out["z_e"], out["z_c"], perm)
    beta = min(1.0, (epoch - 1) * 0.1)
    total = out["loss"] + beta * upper
    keys = optimised_keys(cfg, opt)
    gs = torch.autograd.grad(total, [leaf[k] for k in keys], allow_unused=True)
    P = adam_step(P, dict(zip(keys, gs)), keys, st_vae, lr=opt.vae_lr)
    return P, dict(aprx=aprx.detach(), vae=out["loss"].detach(), upper=upper.detach(), total=total.detach(),
                   z_e=out["z_e"].detach(), z_c=out["z_c"].detach())


# --------------------------------------------------------------------------------------------
# training step (:837-845): zero_grad / backward / Adam
# --------------------------------------------------------------------------------------------


@dataclass
class AdamState:
    steps: Dict[str, int] = field(default_factory=dict)
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)


def adam_step(P, grads, keys, st: AdamState, lr=1e-5, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (:936): no weight decay, no amsgrad; dense over `keys`.
    A tensor whose grad is None (the pair head when the pair loss is replaced by the int 0, :510-511)
    is skipped entirely -- moments and its own step counter untouched -- exactly like torch."""
    for k in keys:
        g = grads.get(k)
        if g is None:
            continue
        st.steps[k] = st.steps.get(k, 0) + 1
        bc1 = 1.0 - b1 ** st.steps[k]
        bc2 = 1.0 - b2 ** st.steps[k]
        if k not in st.m:
            st.m[k] = torch.zeros_like(g)
            st.v[k] = torch.zeros_like(g)
        st.m[k].mul_(b1).add_(g, alpha=1 - b1)
        st.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (st.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        P[k] = P[k] - (lr / bc1) * (st.m[k] / denom)
    return P


def loss_and_grads(P, batch, iteration, cfg, opt, eps_e, eps_c, **kw):
    """Forward + autograd backward on CPU.  Returns (terms, grads for every parameter)."""
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    out = forward_terms(Pg, batch, iteration, cfg, opt, eps_e, eps_c, **kw)
    out["loss"].backward()
    grads = {k: v.grad for k, v in Pg.items()}     # None where the reference leaves .grad None
    return {k: v.detach() for k, v in out.items()}, grads


def train_step(P, batch, iteration, cfg, opt, st: AdamState, eps_e, eps_c, **kw):
    out, grads = loss_and_grads(P, batch, iteration, cfg, opt, eps_e, eps_c, **kw)
    P = adam_step(P, grads, optimised_keys(cfg, opt), st, lr=opt.vae_lr)
    return P, out, grads


# --------------------------------------------------------------------------------------------
# synthetic ECPE-shaped batches; shared by tests and bench
# --------------------------------------------------------------------------------------------

EMO_HIST = np.array([578, 705, 234, 98, 535, 437], dtype=np.float64)


def synthetic_batch(B: int, S: int, cfg: EncoderConfig, V: int, seed: int = 1, shape: str = "A",
                    ) -> Dict[str, torch.Tensor]:
    """Shape-A: dense (mask all ones).  Shape-B: ECPE-like lengths (~77 % padding)."""
    rs = np.random.RandomState(seed)
    lo = 2 if cfg.variant == "roberta" else 1
    ids = rs.randint(lo, cfg.vocab_size, size=(B, S)).astype(np.int64)
    att = np.ones((B, S), dtype=np.int64)
    if shape == "B":
        ln = np.clip(np.round(rs.gamma(shape=6.0, scale=26.9 / 6.0, size=B)), 4, S).astype(np.int64)
        for b in range(B):
            ids[b, ln[b]:] = cfg.pad_id
            att[b, ln[b]:] = 0
    tt = np.zeros((B, S), dtype=np.int64)
    y = (rs.uniform(size=(B, 1)) < 0.5).astype(np.float32)
    if y.sum() == 0:
        y[0, 0] = 1.0
    emo = rs.choice(6, size=(B, 1), p=EMO_HIST / EMO_HIST.sum()).astype(np.int64)
    bow = np.zeros((B, V), dtype=np.float32)
    for b in range(B):
        k = rs.randint(3, 13)
        cols = rs.randint(0, V, size=k)
        np.add.at(bow[b], cols, 1.0)
        bow[b] /= max(bow[b].sum(), 1.0)
    t = torch.from_numpy
    return dict(input_ids=t(ids), attention_masks=t(att), token_type_ids=t(tt), labels=t(y),
                emo_labels=t(emo), cau_labels=t(y.copy()), bow_reps=t(bow))
