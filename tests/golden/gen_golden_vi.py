#!/usr/bin/env python3
"""Golden vectors for the VI / CLUB ablation (drl_classifier_ec_vi.py), produced by EXECUTING the reference's own
`DrlClassifier` class (AST-extracted at run time, nothing copied) on CPU with a locally constructed 2-layer BertModel.

The two-phase update sequence below (aprx optimiser step, CLUB bound with the updated net, beta ramp, main optimiser
step) is the procedure of the reference's training loop (:754-774); every model call in it is reference code.
Weights come from oracle.init_params / init_vi_params (numpy RandomState), so the fixture holds inputs, noise,
permutations and expected outputs only.

    python tests/golden/gen_golden_vi.py         # writes tests/golden/vi_small.npz
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import carel_oracle as O  # noqa: E402
import gen_golden as G  # noqa: E402


def vi_namespace(opt_ns, cfg):
    import math
    import transformers

    def make_bert():
        c = transformers.BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                                    num_attention_heads=cfg.heads, intermediate_size=cfg.intermediate,
                                    max_position_embeddings=cfg.max_pos, type_vocab_size=cfg.type_vocab, layer_norm_eps=cfg.ln_eps,
                                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, hidden_act="gelu")
        c._attn_implementation = "eager"
        return transformers.BertModel(c)

    class _Stub:
        def from_pretrained(self, *a, **k):      # local construction, nothing fetched
            return make_bert()

    ns = dict(torch=torch, nn=nn, math=math, opt=opt_ns, np=np, BertModel=_Stub())
    mod = G.extract(os.path.join(G.REF, "drl_classifier_ec_vi.py"), ["DrlClassifier"])
    exec(compile(mod, "<reference:drl_classifier_ec_vi.py>", "exec"), ns)
    return ns


def main():
    cfg = O.EncoderConfig(layers=2, vocab_size=900)
    opt = O.Opt(pair_bow_dim=211, dropout=0.0, e_num_class=1)
    B, S, wseed, bseed, steps = 16, 128, 31, 41, 3
    epochs = [1, 4, 13]                       # beta = 0, 0.3, 1 (clamped)
    ref_opt = types.SimpleNamespace(bert_dim=768, dropout=0.0, ec_dim=opt.ec_dim, ec_kl_lambda=opt.ec_kl_lambda,
                                    ec_mul_loss_weight=opt.emo_mul_loss_weight, ec_num_class=1, epsilon=opt.epsilon,
                                    kl_ann_iterations=opt.kl_ann_iterations, label_smoothing=opt.label_smoothing,
                                    pair_bow_dim=opt.pair_bow_dim, pair_mul_loss_weight=opt.pair_mul_loss_weight, pair_num_class=1)
    ns = vi_namespace(ref_opt, cfg)
    torch.manual_seed(1234)
    model = ns["DrlClassifier"](ref_opt)
    P = {**O.init_params(cfg, opt, seed=wseed), **O.init_vi_params(opt, seed=wseed + 1)}
    sd = model.state_dict()
    extra = [k for k in sd if k not in P]
    assert all(("position_ids" in k) or ("token_type_ids" in k) for k in extra), extra
    assert not [k for k in P if k not in sd]
    model.load_state_dict({**{k: sd[k] for k in extra}, **P})
    batch = O.synthetic_batch(B, S, cfg, opt.pair_bow_dim, seed=bseed, shape="B")
    batch["emo_labels"] = batch["labels"].clone()          # the VI script's emotion label is binary (one-logit BCE head)
    model.train()
    ec_aprx_params, other_params = model.get_params()
    ec_aprx_opt = torch.optim.Adam(ec_aprx_params, lr=opt.aprx_lr)
    vae_and_cls_opt = torch.optim.Adam(other_params, lr=opt.vae_lr)
    rec = dict(meta=np.array([B, S, cfg.layers, cfg.vocab_size, opt.pair_bow_dim, wseed, bseed, steps], dtype=np.int64),
               epochs=np.array(epochs, dtype=np.int64),
               versions=np.array(f"torch={torch.__version__};transformers={__import__('transformers').__version__}"))
    for k, v in batch.items():
        rec["in_" + k] = v.numpy()
    for s in range(steps):
        torch.manual_seed(2000 + s)
        rec[f"eps_e_{s}"] = torch.randn(opt.ec_dim).numpy()
        rec[f"eps_c_{s}"] = torch.randn(opt.ec_dim).numpy()
        torch.manual_seed(2000 + s)
        e_emb, c_emb, aprx, vae = model(batch["input_ids"], batch["attention_masks"], batch["token_type_ids"],
                                        batch["emo_labels"].view(-1, 1), batch["cau_labels"].view(-1, 1), batch["labels"].view(-1, 1),
                                        batch["bow_reps"], 5 + s)
        rec[f"z_e_{s}"], rec[f"z_c_{s}"] = e_emb.detach().numpy().copy(), c_emb.detach().numpy().copy()
        rec[f"aprx_{s}"], rec[f"vae_{s}"] = np.float64(aprx.item()), np.float64(vae.item())
        ec_aprx_opt.zero_grad()
        aprx.backward(retain_graph=True)
        if s == 0:
            named = dict(model.named_parameters())
            for k in O.VI_KEYS:
                rec["ga_" + k] = named[k].grad.numpy().copy()
        ec_aprx_opt.step()
        torch.manual_seed(3000 + s)
        rec[f"perm_{s}"] = torch.randperm(B).numpy()
        torch.manual_seed(3000 + s)
        rj = model.get_ec_upper_loss(e_emb, c_emb)
        rec[f"upper_{s}"] = np.float64(rj.item())
        if s == 1:       # gradient of the bound wrt the sampled embeddings (what reaches the encoder)
            ge, gc = torch.autograd.grad(rj, [e_emb, c_emb], retain_graph=True)
            rec["dz_e_1"], rec["dz_c_1"] = ge.numpy().copy(), gc.numpy().copy()
        beta = min(1.0, (epochs[s] - 1) * 0.1)
        vae = vae + beta * rj
        rec[f"total_{s}"] = np.float64(vae.item())
        vae_and_cls_opt.zero_grad()
        vae.backward()
        if s == 1:
            named = dict(model.named_parameters())
            for k in G.SLICE_KEYS:
                if k in named and named[k].grad is not None:
                    rec["g_" + k] = G.slices(named[k].grad)
                    rec["gn_" + k] = np.float32(named[k].grad.norm().item())
        vae_and_cls_opt.step()
    named = dict(model.named_parameters())
    for k in list(G.SLICE_KEYS) + list(O.VI_KEYS):
        if k in named:
            rec["w_" + k] = G.slices(named[k]) if k in G.SLICE_KEYS else named[k].detach().numpy().copy()
    np.savez_compressed(os.path.join(G.OUT, "vi_small.npz"), **rec)
    print({k: float(v) for k, v in rec.items() if k.startswith(("aprx_", "vae_", "upper_", "total_"))})


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
