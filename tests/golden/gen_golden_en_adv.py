#!/usr/bin/env python3
"""Golden vectors for the three-space adversarial model (drl_classifier_en.py, config 4), produced by EXECUTING the
reference's own `DrlClassifier` class (AST-extracted at run time, nothing copied) on CPU around a locally constructed
2-layer RobertaModel.

The update sequence below -- five discriminator backward calls with retain_graph, the vae backward, then six Adam
steps -- is the procedure of the reference's training loop (:919-947) with the optimisers its script body builds (RMSprop for
the five discriminators, Adam for the rest, :1056-1062); every model call in it is reference code.
Weights come from oracle.carel_oracle_en.init_params (numpy RandomState), so the fixture holds inputs, noise and expected
outputs only.

    python tests/golden/gen_golden_en_adv.py         # writes tests/golden/en_adv_small.npz
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
from oracle import carel_oracle_en as OE  # noqa: E402
import gen_golden as G  # noqa: E402

HEAD_KEYS = ["content_disc.weight", "content_disc.bias", "emotion_disc.weight", "emotion_disc.bias", "cause_disc.weight",
             "cause_disc.bias", "ec_disc.weight", "ec_disc.bias", "ce_disc.weight", "ce_disc.bias", "content_classifier.weight",
             "content_classifier.bias", "emotion_classifier.weight", "emotion_classifier.bias", "cause_classifier.weight",
             "cause_classifier.bias", "pair_classifier.weight", "pair_classifier.bias", "decoder.weight", "decoder.bias"]
ENC_KEYS = [k for k in G.SLICE_KEYS if k.startswith("encoder.")]


def en_namespace(opt_ns, cfg):
    import math
    import transformers

    def make_roberta():
        c = transformers.RobertaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                                       num_attention_heads=cfg.heads, intermediate_size=cfg.intermediate,
                                       max_position_embeddings=cfg.max_pos, type_vocab_size=cfg.type_vocab, layer_norm_eps=cfg.ln_eps,
                                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, hidden_act="gelu", pad_token_id=cfg.pad_id)
        c._attn_implementation = "eager"
        return transformers.RobertaModel(c)

    class _Stub:
        def from_pretrained(self, *a, **k):      # local construction, nothing fetched
            return make_roberta()

    ns = dict(torch=torch, nn=nn, math=math, opt=opt_ns, np=np, RobertaModel=_Stub())
    mod = G.extract(os.path.join(G.REF, "drl_classifier_en.py"), ["DrlClassifier"])
    exec(compile(mod, "<reference:drl_classifier_en.py>", "exec"), ns)
    return ns


def main():
    cfg = O.EncoderConfig(layers=2, vocab_size=900, max_pos=514, type_vocab=1, ln_eps=1e-5, variant="roberta", pad_id=1)
    opt = OE.OptEn(pair_bow_dim=211, dropout=0.0)
    B, S, wseed, bseed, steps = 16, 128, 51, 61, 3
    ref_opt = types.SimpleNamespace(**vars(opt))
    ns = en_namespace(ref_opt, cfg)
    torch.manual_seed(1234)
    model = ns["DrlClassifier"](ref_opt)
    P = OE.init_params(cfg, opt, seed=wseed)
    sd = model.state_dict()
    extra = [k for k in sd if k not in P]
    assert all(("position_ids" in k) or ("token_type_ids" in k) for k in extra), extra
    assert not [k for k in P if k not in sd]
    heads = lambda ks: [k for k in ks if not k.startswith("encoder.")]     # noqa: E731
    assert heads(k for k in sd if k in P) == heads(P), "registration order differs from oracle.param_shapes"
    model.load_state_dict({**{k: sd[k] for k in extra}, **P})
    batch = OE.synthetic_batch(B, S, cfg, opt.pair_bow_dim, seed=bseed, shape="B")
    model.train()
    groups = model.get_params()
    named = dict(model.named_parameters())
    ids = {id(p): k for k, p in named.items()}
    groups = [list(g) for g in groups]
    okeys = OE.group_keys(cfg, opt)
    got = [[ids[id(p)] for p in g] for g in groups]          # encoder keys: HF's own order, compared as a set
    assert [heads(g) for g in got] == [heads(g) for g in okeys] and [set(g) for g in got] == [set(g) for g in okeys], \
        "get_params grouping differs from oracle.group_keys"
    opts = [torch.optim.RMSprop(g, lr=opt.adv_lr) for g in groups[:5]] + [torch.optim.Adam(groups[5], lr=opt.vae_lr)]    # :1056-1062
    rec = dict(meta=np.array([B, S, cfg.layers, cfg.vocab_size, opt.pair_bow_dim, wseed, bseed, steps], dtype=np.int64),
               versions=np.array(f"torch={torch.__version__};transformers={__import__('transformers').__version__}"))
    for k, v in batch.items():
        rec["in_" + k] = v.numpy()
    for s in range(steps):
        torch.manual_seed(4000 + s)                       # sample_prior order: content, emotion, cause (:238-240)
        rec[f"eps_con_{s}"] = torch.randn(opt.con_dim).numpy()
        rec[f"eps_e_{s}"] = torch.randn(opt.ec_dim).numpy()
        rec[f"eps_c_{s}"] = torch.randn(opt.ec_dim).numpy()
        torch.manual_seed(4000 + s)
        losses = model(batch["input_ids"], batch["attention_masks"], batch["token_type_ids"], batch["emo_labels"].view(-1, 1),
                       batch["cau_labels"].view(-1, 1), batch["labels"].view(-1, 1), batch["bow_reps"], 7 + s)
        rec[f"losses_{s}"] = np.array([float(v.item()) for v in losses], dtype=np.float64)
        cd_e, cd_c, ed, ecd, cad, ced, vae = losses
        opts[0].zero_grad(); (cd_e + cd_c).backward(retain_graph=True)        # noqa: E702   :919-937, same order
        opts[1].zero_grad(); ed.backward(retain_graph=True)                  # noqa: E702
        opts[3].zero_grad(); ecd.backward(retain_graph=True)                 # noqa: E702
        opts[2].zero_grad(); cad.backward(retain_graph=True)                 # noqa: E702
        opts[4].zero_grad(); ced.backward(retain_graph=True)                 # noqa: E702
        opts[5].zero_grad(); vae.backward()                                  # noqa: E702
        if s == 1:
            for k in HEAD_KEYS + ENC_KEYS:
                if named[k].grad is not None:
                    rec["g_" + k] = G.slices(named[k].grad)
                    rec["gn_" + k] = np.float32(named[k].grad.norm().item())
        for o in opts:
            o.step()
    for k in HEAD_KEYS + ENC_KEYS:
        rec["w_" + k] = G.slices(named[k])
    # get_pair_preds (:336-353): raw logits, emotion noise before cause noise
    model.eval()
    torch.manual_seed(4100)
    rec["pp_eps_e"], rec["pp_eps_c"] = torch.randn(opt.ec_dim).numpy(), torch.randn(opt.ec_dim).numpy()
    torch.manual_seed(4100)
    with torch.no_grad():
        rec["pp_logits"] = model.get_pair_preds(batch["input_ids"], batch["attention_masks"], batch["token_type_ids"]).numpy()
    np.savez_compressed(os.path.join(G.OUT, "en_adv_small.npz"), **rec)
    print({k: v for k, v in rec.items() if k.startswith("losses_")})


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
