"""CPU restatement of the three-space adversarial model of `drl_classifier_en.py` (config 4).

TEST INFRASTRUCTURE ONLY, like carel_oracle.py: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product path (carel_vae_amd/) never imports it.

Pinned by tests/golden/en_adv_small.npz, which tests/golden/gen_golden_en_adv.py produced by EXECUTING the reference's
own `DrlClassifier` class of drl_classifier_en.py (AST-extracted at generation time, nothing copied) around a locally
constructed RobertaModel.  The encoder is carel_oracle.encoder_forward with the RoBERTa geometry.

Line numbers below are those of /root/reference/drl_classifier_en.py.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import carel_oracle as O


@dataclass
class OptEn:
    """argparse namespace of drl_classifier_en.py :29-61, defaults identical."""
    max_len: int = 128
    ec_num_class: int = 1
    pair_num_class: int = 1
    ec_dim: int = 24
    con_dim: int = 384
    pair_bow_dim: int = 23771
    bert_dim: int = 768
    kl_ann_iterations: int = 20000
    epochs: int = 10
    batch_size: int = 64
    ec_kl_lambda: float = 0.03
    con_kl_lambda: float = 0.03
    label_smoothing: float = 0.1
    con_adv_loss_weight: float = 0.03
    ec_adv_loss_weight: float = 1.0
    ecce_adv_loss_weight: float = 3.0
    con_mul_loss_weight: float = 3.0
    ec_mul_loss_weight: float = 10.0
    pair_mul_loss_weight: float = 30.0
    dropout: float = 0.5
    epsilon: float = 1e-8
    adv_lr: float = 0.001
    vae_lr: float = 1e-5
    language: str = "en"
    model_id: str = "oracle-en"


# dropout sites of the ten nn.Dropout calls in forward (:248-283), in call order
SITE_CDISC_E, SITE_CDISC_C, SITE_CMUL, SITE_EDISC, SITE_ECDISC, SITE_EMUL, SITE_CAUDISC, SITE_CEDISC, SITE_CAUMUL, SITE_PAIR = range(110, 120)

LOSS_NAMES = ("content_disc_emo", "content_disc_cau", "emotion_disc", "ec_disc", "cause_disc", "ce_disc", "vae")
LATENT_HEADS = ("content_mu", "content_log_var", "emotion_mu", "emotion_log_var", "cause_mu", "cause_log_var")
DISC_GROUPS = ("content_disc", "emotion_disc", "cause_disc", "ec_disc", "ce_disc")       # get_params order (:365-369)
OTHER_HEADS = ("decoder", "emotion_classifier", "cause_classifier", "pair_classifier", "content_classifier")   # :370-375


def param_shapes(cfg: O.EncoderConfig, opt: OptEn) -> Dict[str, tuple]:
    """state_dict key -> shape in registration order (:157-203)."""
    base = O.param_shapes(cfg, O.Opt(ec_dim=opt.ec_dim, bert_dim=opt.bert_dim, pair_bow_dim=opt.pair_bow_dim))
    s = {k: v for k, v in base.items() if k.startswith("encoder.")}
    H, D, Cd, V, E = opt.bert_dim, opt.ec_dim, opt.con_dim, opt.pair_bow_dim, opt.ec_num_class

    def lin(name, out, inp):
        s[name + ".weight"] = (out, inp)
        s[name + ".bias"] = (out,)
    lin("content_mu", Cd, H); lin("content_log_var", Cd, H)
    lin("emotion_mu", D, H); lin("emotion_log_var", D, H)
    lin("cause_mu", D, H); lin("cause_log_var", D, H)
    lin("emotion_disc", E, Cd); lin("content_disc", V, D); lin("cause_disc", E, Cd); lin("ec_disc", E, D); lin("ce_disc", E, D)
    lin("content_classifier", V, Cd); lin("emotion_classifier", E, D); lin("cause_classifier", E, D)
    lin("pair_classifier", opt.pair_num_class, 2 * D)
    lin("decoder", V, 2 * D + Cd)
    return s


def group_keys(cfg: O.EncoderConfig, opt: OptEn):
    """The six parameter groups of get_params() (:357-376), each a list of keys."""
    keys = list(param_shapes(cfg, opt).keys())
    groups = [[k for k in keys if k.startswith(g + ".")] for g in DISC_GROUPS]
    other = [k for k in keys if k.startswith("encoder.")]
    for h in OTHER_HEADS:
        other += [k for k in keys if k.startswith(h + ".")]
    return groups + [other]


def init_params(cfg: O.EncoderConfig, opt: OptEn, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Same recipe as carel_oracle.init_params (numpy RandomState stream)."""
    rs = np.random.RandomState(seed)
    shapes = param_shapes(cfg, opt)
    out: Dict[str, torch.Tensor] = {}
    for k, shp in shapes.items():
        if k.startswith("encoder."):
            a = (1.0 + 0.05 * rs.standard_normal(shp)) if "LayerNorm.weight" in k else 0.02 * rs.standard_normal(shp)
        else:
            fan_in = shp[1] if len(shp) == 2 else shapes[k.replace(".bias", ".weight")][1]
            bound = 1.0 / math.sqrt(fan_in)
            a = rs.uniform(-bound, bound, size=shp)
        out[k] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return out


def kl_anneal_weight(iteration: int, opt: OptEn, lam: float) -> float:
    """:605-613"""
    return (math.tanh((iteration - opt.kl_ann_iterations * 1.5) / (opt.kl_ann_iterations / 3)) + 1) * lam


def tail_forward(P, pooled, emo_labels, cau_labels, pair_labels, bow, iteration: int, opt: OptEn, eps: Dict[str, torch.Tensor],
                 train: bool = False, seed: Optional[int] = None, row_offset: int = 0, global_label_sum: Optional[float] = None,
                 global_n: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Everything after pooler_output (:220-334).  eps: {"con": [con_dim], "e": [ec_dim], "c": [ec_dim]} -- sample_prior
    is called for content, emotion, cause in that order (:238-240), one vector per call shared by the batch (:417-423).
    Labels are float [B, 1] (the `_en` dataset yields float emotion labels, :132)."""
    B = pooled.shape[0]
    D, Cd, V, ls = opt.ec_dim, opt.con_dim, opt.pair_bow_dim, opt.label_smoothing

    def lin(x, name):
        return x @ P[name + ".weight"].t() + P[name + ".bias"]
    mu_con, lv_con = lin(pooled, "content_mu"), lin(pooled, "content_log_var")
    mu_e, lv_e = lin(pooled, "emotion_mu"), lin(pooled, "emotion_log_var")
    mu_c, lv_c = lin(pooled, "cause_mu"), lin(pooled, "cause_log_var")
    z_con = mu_con + eps["con"] * torch.exp(lv_con)
    z_e = mu_e + eps["e"] * torch.exp(lv_e)
    z_c = mu_c + eps["c"] * torch.exp(lv_c)
    gen = torch.cat((z_e, z_c, z_con), dim=1)          # :243
    pair_emb = torch.cat((z_e, z_c), dim=1)            # :244
    pd = opt.dropout if train else 0.0

    def drop(t, site):
        m = O.dropout_scale_mask(seed, site, tuple(t.shape), pd, row_offset)
        return t if m is None else t * m
    emo_y = emo_labels.view(B, -1).to(torch.float32)
    cau_y = cau_labels.view(B, -1).to(torch.float32)
    pair_y = pair_labels.view(B, -1).to(torch.float32)
    bow_t = bow * (1 - ls) + ls / V                      # :447, :455, :548
    ec_t = lambda y: y * (1 - ls) + ls / opt.ec_num_class     # noqa: E731   :526, :565, :581

    def entropy(p):                                      # :531-536
        return (p * torch.log(p + opt.epsilon)).sum(dim=1).mean()
    # content space (:246-256)
    p_cd_e = torch.softmax(lin(drop(z_e.detach(), SITE_CDISC_E), "content_disc"), dim=1)
    p_cd_c = torch.softmax(lin(drop(z_c.detach(), SITE_CDISC_C), "content_disc"), dim=1)
    cd_e, cd_c = O.bce_prob(p_cd_e, bow_t).mean(), O.bce_prob(p_cd_c, bow_t).mean()
    cent_e, cent_c = entropy(p_cd_e), entropy(p_cd_c)
    con_mul = O.bce_prob(torch.softmax(lin(drop(z_con, SITE_CMUL), "content_classifier"), dim=1), bow_t).mean()
    # emotion space (:258-268)
    p_ed = torch.sigmoid(lin(drop(z_con.detach(), SITE_EDISC), "emotion_disc"))
    p_ec = torch.sigmoid(lin(drop(z_c.detach(), SITE_ECDISC), "ec_disc"))
    ed, ecd = O.bce_prob(p_ed, ec_t(emo_y)).mean(), O.bce_prob(p_ec, ec_t(emo_y)).mean()
    ent_ed, ent_ec = entropy(p_ed), entropy(p_ec)
    emo_mul = O.bce_prob(torch.sigmoid(lin(drop(z_e, SITE_EMUL), "emotion_classifier")), ec_t(emo_y)).mean()
    # cause space (:270-280)
    p_cad = torch.sigmoid(lin(drop(z_con.detach(), SITE_CAUDISC), "cause_disc"))
    p_ce = torch.sigmoid(lin(drop(z_e.detach(), SITE_CEDISC), "ce_disc"))
    cad, ced = O.bce_prob(p_cad, ec_t(cau_y)).mean(), O.bce_prob(p_ce, ec_t(cau_y)).mean()
    ent_cad, ent_ce = entropy(p_cad), entropy(p_ce)
    cau_mul = O.bce_prob(torch.sigmoid(lin(drop(z_c, SITE_CAUMUL), "cause_classifier")), ec_t(cau_y)).mean()
    # pair (:283, :587-603): no infinity guard in this script
    xp = lin(drop(pair_emb, SITE_PAIR), "pair_classifier")
    sy = pair_y.sum() if global_label_sum is None else torch.tensor(float(global_label_sum))       # data-parallel shard: global batch
    pw = ((B if global_n is None else global_n) - sy) / sy
    pair = O.bce_logits_posw(xp, ec_t(pair_y), pw).mean()
    # KL (:285-303)

    def kl(mu, lv):
        return (-0.5 * (1 + lv - lv.exp() - mu.pow(2)).sum(dim=1)).mean()
    kl_e, kl_c, kl_con = kl(mu_e, lv_e), kl(mu_c, lv_c), kl(mu_con, lv_con)
    if iteration < opt.kl_ann_iterations:
        kl_e = kl_anneal_weight(iteration, opt, opt.ec_kl_lambda) * kl_e
        kl_c = kl_anneal_weight(iteration, opt, opt.ec_kl_lambda) * kl_c
        kl_con = kl_anneal_weight(iteration, opt, opt.con_kl_lambda) * kl_con
    rec = O.bce_prob(torch.softmax(lin(gen, "decoder"), dim=1), bow_t).mean()       # :306-307
    vae = (opt.con_adv_loss_weight * (cent_e + cent_c) + opt.ec_adv_loss_weight * (ent_ed + ent_cad)
           + opt.ecce_adv_loss_weight * (ent_ec + ent_ce) + opt.ec_mul_loss_weight * (emo_mul + cau_mul)
           + opt.con_mul_loss_weight * con_mul + opt.pair_mul_loss_weight * pair + kl_e + kl_c + kl_con + rec)     # :325-332
    return dict(content_disc_emo=cd_e, content_disc_cau=cd_c, emotion_disc=ed, ec_disc=ecd, cause_disc=cad, ce_disc=ced, vae=vae,
                cent_e=cent_e, cent_c=cent_c, ent_ed=ent_ed, ent_cad=ent_cad, ent_ec=ent_ec, ent_ce=ent_ce, emo_mul=emo_mul,
                cau_mul=cau_mul, con_mul=con_mul, pair=pair, kl_e=kl_e, kl_c=kl_c, kl_con=kl_con, rec=rec,
                mu_e=mu_e, mu_c=mu_c, mu_con=mu_con, lv_e=lv_e, lv_c=lv_c, lv_con=lv_con, z=gen, pair_logit=xp)


def forward_terms(P, batch, iteration, cfg, opt: OptEn, eps, train=False, seed=None, quant: O.Quant = None, row_offset=0, **kw):
    """`DrlClassifier.forward` (:205-334) with every term exposed."""
    pooled = O.encoder_forward(P, batch["input_ids"], batch["attention_masks"], batch["token_type_ids"], cfg,
                               train=train, seed=seed, row_offset=row_offset, quant=quant)
    out = tail_forward(P, pooled, batch["emo_labels"], batch["cau_labels"], batch["labels"], batch["bow_reps"], iteration, opt, eps,
                       train=train, seed=seed, row_offset=row_offset, **kw)
    out["pooled"] = pooled
    return out


def pair_logits(P, ids, att, tt, cfg, opt: OptEn, eps_e, eps_c, quant: O.Quant = None):
    """get_pair_preds (:336-353): raw logits of the pair head; emotion noise drawn before cause noise; no content sample."""
    pooled = O.encoder_forward(P, ids, att, tt, cfg, quant=quant)
    lin = lambda n: pooled @ P[n + ".weight"].t() + P[n + ".bias"]     # noqa: E731
    z = torch.cat((lin("emotion_mu") + eps_e * lin("emotion_log_var").exp(), lin("cause_mu") + eps_c * lin("cause_log_var").exp()), dim=1)
    return z @ P["pair_classifier.weight"].t() + P["pair_classifier.bias"]


def loss_and_grads(P, batch, iteration, cfg, opt: OptEn, eps, **kw):
    """The six backward calls of the step (:919-939) and the gradients each optimiser then sees.  Every discriminator
    input is detached, so the six discriminator losses reach discriminator parameters only; the entropy terms inside the
    vae loss ALSO reach only discriminator parameters (their predictions come from detached embeddings), and because
    each zero_grad clears just its own group, those gradients add to the discriminator losses' before the steps."""
    leaf = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    out = forward_terms(leaf, batch, iteration, cfg, opt, eps, **kw)
    groups = group_keys(cfg, opt)
    disc_losses = [out["content_disc_emo"] + out["content_disc_cau"], out["emotion_disc"], out["cause_disc"], out["ec_disc"], out["ce_disc"]]
    grads: Dict[str, torch.Tensor] = {}
    for keys, loss in zip(groups[:5], disc_losses):
        for k, g in zip(keys, torch.autograd.grad(loss, [leaf[k] for k in keys], retain_graph=True)):
            grads[k] = g
    allk = [k for g in groups for k in g]
    for k, g in zip(allk, torch.autograd.grad(out["vae"], [leaf[k] for k in allk], allow_unused=True)):
        if g is not None:
            grads[k] = grads[k] + g if k in grads else g
    return {k: v.detach() for k, v in out.items()}, grads


def rmsprop_step(P, grads, keys, st: "O.AdamState", lr, alpha=0.99, eps=1e-8):
    """torch.optim.RMSprop defaults (no momentum, not centered, no weight decay): v = alpha v + (1 - alpha) g^2;
    p -= lr g / (sqrt(v) + eps).  The running average lives in st.v."""
    for k in keys:
        g = grads.get(k)
        if g is None:
            continue
        if k not in st.v:
            st.v[k] = torch.zeros_like(g)
        st.v[k].mul_(alpha).addcmul_(g, g, value=1 - alpha)
        P[k] = P[k] - lr * (g / (st.v[k].sqrt() + eps))
    return P


def train_step(P, batch, iteration, cfg, opt: OptEn, states, eps, **kw):
    """One iteration of the training loop (:904-947) with the optimisers the script builds (:1056-1062): RMSprop(adv_lr)
    for the five discriminators, Adam(vae_lr) for the rest.  states: six carel_oracle.AdamState, in get_params order."""
    out, grads = loss_and_grads(P, batch, iteration, cfg, opt, eps, **kw)
    P = dict(P)
    for i, keys in enumerate(group_keys(cfg, opt)):
        if i < 5:
            P = rmsprop_step(P, grads, keys, states[i], lr=opt.adv_lr)
        else:
            P = O.adam_step(P, grads, keys, states[i], lr=opt.vae_lr)
    return P, out, grads


def synthetic_batch(B, S, cfg, V, seed=1, shape="A"):
    """carel_oracle.synthetic_batch with the binary float emotion label of the `_en` dataset (:132)."""
    b = O.synthetic_batch(B, S, cfg, V, seed=seed, shape=shape)
    rs = np.random.RandomState(seed + 7919)
    b["emo_labels"] = torch.from_numpy((rs.uniform(size=(B, 1)) < 0.5).astype(np.float32))
    return b
