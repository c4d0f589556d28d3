"""CPU oracle (TEST INFRASTRUCTURE, fp32 PyTorch) of the sentence-embedding fine-tune of chi_ec_sentence_transformer.py /
en_ec_sentence_transformer.py: `SentenceTransformer(name)` (:22), `losses.BatchSemiHardTripletLoss(model, margin)` (:78),
`model.fit(train_objectives=[(loader, loss)], epochs, warmup_steps, output_path)` (:84-87).

PARITY UNPINNED.  Every line of arithmetic on this path lives in the third-party package `sentence_transformers`, which
is neither installed in this container nor vendored under /root/reference, and whose version the reference does not pin
(no requirements file).  The reference holds no outputs, fixtures or tests for it.  What follows restates the package's
PUBLISHED algorithm (sentence-transformers 2.x): models.Transformer + models.Pooling(pooling_mode="mean"),
losses.BatchSemiHardTripletLoss with BatchHardTripletLossDistanceFunction.eucledian_distance, and the loop of
SentenceTransformer.fit (AdamW lr 2e-5 weight_decay 0.01 without decay on biases / LayerNorm, scheduler "WarmupLinear",
max_grad_norm 1).  `triplet_loss_by_definition` is an independent loop-form statement of the same loss that
tests/test_oracle_triplet.py holds the vectorised restatement to.  Only tests/ may import this module.
"""
from typing import Dict, List

import torch

from . import carel_oracle as O


def mean_pool(hidden: torch.Tensor, att_mask: torch.Tensor) -> torch.Tensor:
    """models.Pooling, mode "mean": sum of the attended token states / clamp(number of attended tokens, 1e-9)."""
    m = att_mask.to(hidden.dtype).unsqueeze(-1)
    return (hidden * m).sum(1) / m.sum(1).clamp(min=1e-9)


def encode(P: Dict[str, torch.Tensor], ids, att_mask, token_type, cfg: O.EncoderConfig, train=False, seed=None, quant=None,
           normalize=None) -> torch.Tensor:
    """sentence embedding = mean pooling of the LAST hidden states (the BERT pooler is not used by models.Transformer);
    normalize (default: the MPNet variant only): models.Normalize, the third module of all-mpnet-base-v2 -- F.normalize(p=2, dim=1)."""
    taps = {}
    O.encoder_forward(P, ids, att_mask, token_type, cfg, train=train, seed=seed, quant=quant, taps=taps)
    emb = mean_pool(taps[f"x{cfg.layers}"], att_mask)
    if normalize is None:
        normalize = cfg.variant == "mpnet"
    return torch.nn.functional.normalize(emb, p=2, dim=1) if normalize else emb


def mpnet_key_to_internal(k: str) -> str:
    """transformers MPNetModel state-dict key -> the BERT-style key this package (and the oracle) stores it under."""
    k = k.replace(".attention.attn.q.", ".attention.self.query.").replace(".attention.attn.k.", ".attention.self.key.")
    k = k.replace(".attention.attn.v.", ".attention.self.value.").replace(".attention.attn.o.", ".attention.output.dense.")
    k = k.replace(".attention.LayerNorm.", ".attention.output.LayerNorm.")
    return k


def params_from_mpnet_state_dict(sd: Dict[str, torch.Tensor], cfg: O.EncoderConfig) -> Dict[str, torch.Tensor]:
    """HF MPNetModel weights -> oracle parameter dict (keys "encoder." + internal name; zero token-type placeholder row)."""
    P = {"encoder." + mpnet_key_to_internal(k): v.detach().clone().float() for k, v in sd.items() if not k.endswith("position_ids")}
    P["encoder.embeddings.token_type_embeddings.weight"] = torch.zeros((1, cfg.hidden))
    if "encoder.pooler.dense.weight" not in P:
        P["encoder.pooler.dense.weight"] = torch.zeros((cfg.hidden, cfg.hidden))
        P["encoder.pooler.dense.bias"] = torch.zeros(cfg.hidden)
    return P


def euclidean_distance(emb: torch.Tensor) -> torch.Tensor:
    dot = emb @ emb.t()
    sq = torch.diag(dot)
    d = sq.unsqueeze(0) - 2.0 * dot + sq.unsqueeze(1)
    d = torch.where(d < 0, torch.zeros_like(d), d)
    mask = d.eq(0).to(d.dtype)
    return (1.0 - mask) * torch.sqrt(d + mask * 1e-16)


def _masked_minimum(data, mask, dim=1):
    axis_max = data.max(dim, keepdim=True)[0]
    return ((data - axis_max) * mask).min(dim, keepdim=True)[0] + axis_max


def _masked_maximum(data, mask, dim=1):
    axis_min = data.min(dim, keepdim=True)[0]
    return ((data - axis_min) * mask).max(dim, keepdim=True)[0] + axis_min


def batch_semi_hard_triplet_loss(labels: torch.Tensor, emb: torch.Tensor, margin: float) -> torch.Tensor:
    """losses.BatchSemiHardTripletLoss.batch_semi_hard_triplet_loss (the TensorFlow-addons formulation)."""
    labels = labels.reshape(-1, 1)
    pd = euclidean_distance(emb)
    adj = labels == labels.t()
    adj_not = ~adj
    B = labels.numel()
    tile = pd.repeat([B, 1])
    mask = adj_not.repeat([B, 1]) & (tile > pd.t().reshape(-1, 1))
    mask_final = (mask.to(pd.dtype).sum(1, keepdim=True) > 0.0).reshape(B, B).t()
    neg_outside = _masked_minimum(tile, mask.to(pd.dtype)).reshape(B, B).t()
    neg_inside = _masked_maximum(pd, adj_not.to(pd.dtype)).repeat([1, B])
    semi_hard = torch.where(mask_final, neg_outside, neg_inside)
    loss_mat = (pd - semi_hard) + margin
    mask_pos = adj.to(pd.dtype) - torch.eye(B, dtype=pd.dtype)
    num_pos = mask_pos.sum()
    return torch.clamp(loss_mat * mask_pos, min=0.0).sum() / num_pos


def triplet_loss_by_definition(labels: torch.Tensor, emb: torch.Tensor, margin: float) -> torch.Tensor:
    """The same loss, written as loops from its definition: for every anchor a and positive p != a of the same label, the
    negative distance is the smallest D[a][n] among negatives farther than the positive, else the largest D[a][n] among
    all negatives (else D's row minimum, 0); mean over the positive pairs of max(D[a][p] - Dneg + margin, 0)."""
    D = euclidean_distance(emb)
    lab = labels.reshape(-1).tolist()
    B = len(lab)
    total, npos = emb.new_zeros(()), 0
    for a in range(B):
        for p in range(B):
            if a == p or lab[a] != lab[p]:
                continue
            npos += 1
            negs = [n for n in range(B) if lab[n] != lab[a]]
            outside = [n for n in negs if float(D[a, n]) > float(D[a, p])]
            if outside:
                dneg = min((D[a, n] for n in outside), key=float)
            elif negs:
                dneg = max((D[a, n] for n in negs), key=float)
            else:
                dneg = D[a].min()
            total = total + torch.clamp(D[a, p] - dneg + margin, min=0.0)
    return total / npos


NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def encoder_keys(P):
    """the parameters models.Transformer holds and uses: the whole BertModel except its pooler (no gradient -> never stepped)"""
    return [k for k in P if k.startswith("encoder.") and not k.startswith("encoder.pooler.")
            and not (k.endswith("token_type_embeddings.weight") and "encoder.encoder.relative_attention_bias.weight" in P)]   # MPNet: no token types


def fit_steps(P: Dict[str, torch.Tensor], batches: List[dict], cfg: O.EncoderConfig, margin: float, lr=2e-5, weight_decay=0.01,
              warmup_steps=0, total_steps=None, max_grad_norm=1.0, train=False, seeds=None, quant=None):
    """The loop of SentenceTransformer.fit over the given batches (dicts with input_ids, attention_masks, token_type_ids,
    labels): loss -> backward -> clip_grad_norm_ -> AdamW.step -> zero_grad -> WarmupLinear.step.  Returns the per-step
    losses, the per-step gradient norms and the updated parameters (new dict)."""
    keys = encoder_keys(P)
    W = {k: (v.clone().requires_grad_(True) if k in keys else v.clone()) for k, v in P.items()}
    decay = [W[k] for k in keys if not any(nd in k for nd in NO_DECAY)]
    no_decay = [W[k] for k in keys if any(nd in k for nd in NO_DECAY)]
    optim = torch.optim.AdamW([{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0.0}], lr=lr)
    total = total_steps if total_steps is not None else len(batches)

    def lr_lambda(step):                                    # transformers.get_linear_schedule_with_warmup
        if step < warmup_steps:
            return float(step) / float(max(1, warmup_steps))
        return max(0.0, float(total - step) / float(max(1, total - warmup_steps)))
    sched = torch.optim.lr_scheduler.LambdaLR(optim, lr_lambda)
    losses, norms = [], []
    for i, b in enumerate(batches):
        emb = encode(W, b["input_ids"], b["attention_masks"], b["token_type_ids"], cfg, train=train, seed=None if seeds is None else seeds[i], quant=quant)
        loss = batch_semi_hard_triplet_loss(b["labels"], emb, margin)
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_([W[k] for k in keys], max_grad_norm)))
        optim.step()
        optim.zero_grad()
        sched.step()
        losses.append(float(loss.detach()))
    return losses, norms, {k: v.detach().clone() for k, v in W.items()}
