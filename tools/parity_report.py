#!/usr/bin/env python3
"""Measured bf16-vs-fp32 parity of the HIP forward on every golden case and on the bench configuration (B = 64, S = 128, 12
layers, vocab 21 128, V = 23 771, dropout off; reference there = the CPU oracle in fp32, which tests/test_oracle_golden.py
holds to the reference's own outputs).  Prints what the tolerances in tests/test_gpu_model.py are set from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import test_gpu_model as T
from oracle import carel_oracle as O
from carel_vae_amd import drl_classifier as M

gd = os.path.join(os.path.dirname(T.__file__), "golden")
W = dict(mmd=30.0, emo=10.0, cau=10.0, pair=30.0, kl_e=1.0, kl_c=1.0, rec=1.0)
def report(name, out, ref):
    terms = {k: abs(float(out[k]) - float(ref[k])) / max(abs(float(ref[k])), 1e-12) for k in T.TERMS}
    lat = {k: T.relnorm(out[k], ref[k]) for k in ("pooled", "mu_e", "lv_e", "mu_c", "lv_c")}
    scale = sum(abs(W[k] * float(ref[k])) for k in T.TERMS)
    dl = abs(float(out["loss"]) - float(ref["loss"]))
    print("%-10s loss %.5f ref %.5f rel %.2e  /scale %.2e | terms %s | latents %s" % (
        name, float(out["loss"]), float(ref["loss"]), dl / abs(float(ref["loss"])), dl / scale,
        " ".join("%s %.1e" % (k, v) for k, v in terms.items()), " ".join("%s %.1e" % (k, v) for k, v in lat.items())), flush=True)
for name, (cfg, opt) in T.CASES.items():
    z, batch = T.load(gd, name)
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    model, P = T.build(cfg, opt, wseed)
    model.train()
    model.set_noise(torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"]))
    out = model.forward_terms(*T.call(model, batch, it0))
    ref = {k: torch.from_numpy(z["t_" + k]) for k in T.TERMS}
    ref.update({k: torch.from_numpy(z[k]) for k in ("pooled", "mu_e", "lv_e", "mu_c", "lv_c")})
    ref["loss"] = torch.from_numpy(z["t_loss"]) if "t_loss" in z.files else sum(W[k] * ref[k] * (-1 if k == "mmd" else 1) for k in T.TERMS)
    report(name, out, ref)
    del model
# bench configuration
cfg, opt = O.EncoderConfig(), O.Opt(dropout=0.0)
torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
for shape in ("A", "B"):
    model, P = T.build(cfg, opt, 0)
    model.train()
    batch = O.synthetic_batch(64, 128, cfg, opt.pair_bow_dim, seed=1, shape=shape)
    g = torch.Generator().manual_seed(3)
    eps_e, eps_c = torch.randn(opt.ec_dim, generator=g), torch.randn(opt.ec_dim, generator=g)
    model.set_noise(eps_e, eps_c)
    out = model.forward_terms(*T.call(model, batch, 3))
    ref = O.forward_terms(P, batch, 3, cfg, opt, eps_e, eps_c)
    report("bench64" + shape, out, ref)
    del model
