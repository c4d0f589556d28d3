#!/usr/bin/env python3
"""Stress test of the stream logic: the same 12-layer model with and without the side streams (weight-gradient stream,
two forward chains, per-layer Adam on the auxiliary stream), many forward/backward passes back to back on dense and
packed batches of different sizes; every encoder-layer gradient must be bit-identical in every pass, and the updated
weights after the first optimiser step too.  python tools/stress_streams.py [passes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import drl_classifier as M
from oracle import carel_oracle as O

def run(passes, verbose=True, layers=12):
    dev = "cuda"
    opt = M.make_opt(pair_bow_dim=2000)
    cfg = M.encoder_config("zh", vocab_size=3000, layers=layers)
    ocfg = O.EncoderConfig(vocab_size=3000)
    models = []
    for overlap in (False, True):
        torch.manual_seed(0)
        m = M.DrlClassifier(opt, cfg, seed=3).to(dev)
        m.train()
        m.overlap_wgrad = overlap
        m.forward_chains = overlap          # (opt-in since round 2: keep the two-chain forward under test)
        models.append((m, M.FusedAdam(m, lr=1e-5, fuse_into_backward=overlap)))
    bad = 0
    for it in range(passes):
        B = (64, 16, 32, 8, 48)[it % 5]
        shape = "AB"[it % 2]
        b = {k: v.to(dev) for k, v in O.synthetic_batch(B, 128, ocfg, opt.pair_bow_dim, seed=100 + it, shape=shape).items()}
        eps = (torch.randn(24), torch.randn(24))
        res = []
        for m, optim in models:
            m.set_noise(*eps)
            loss = m(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], it)
            for p in m.parameters():
                p.grad = None
            loss.backward()
            torch.cuda.synchronize()
            res.append((float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters() if k.startswith("encoder.encoder.layer.")}))
            if m._adam_hook is not None:
                m._adam_hook._join()              # the hook has updated the layers during backward: restored below
        # keep the two replicas identical: copy the plain model's weights into the overlapped one (its layers moved)
        with torch.no_grad():
            models[1][0]._flat.copy_(models[0][0]._flat)
            models[1][0]._shadow_versions = None
            models[1][1]._done = []
        ok = res[0][0] == res[1][0] and all(torch.equal(res[0][1][k], res[1][1][k]) for k in res[0][1])
        bad += (not ok)
        if verbose:
            print("pass %2d B=%2d shape %s loss %.6f %s" % (it, B, shape, res[0][0], "ok" if ok else "MISMATCH"), flush=True)
            if not ok:
                for k in res[0][1]:
                    if not torch.equal(res[0][1][k], res[1][1][k]):
                        d = (res[0][1][k] - res[1][1][k]).abs()
                        print("    %-60s %8d of %8d differ, max %.3e (|g| max %.3e)" % (k, int((d > 0).sum()), d.numel(), float(d.max()), float(res[0][1][k].abs().max())), flush=True)
    return bad


if __name__ == "__main__":
    n_bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
    print("mismatches:", n_bad)
    sys.exit(1 if n_bad else 0)
