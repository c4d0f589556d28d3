#!/usr/bin/env python3
"""(round 4) The first steps after a device-wide synchronise are slow (bench.py: 7.9, 10.8, 8.4 ms, then 7.7).  Where?  Host enqueue time
and GPU time of each of the first steps after a sync, resident inputs, with and without a short pause after the sync."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from carel_vae_amd import drl_classifier as M
from carel_vae_amd.data import synthetic_ecpe_batch
dev = torch.device("cuda", 0)
L.check(L.load().carel_init(0))
opt, cfg = M.make_opt(), M.encoder_config("zh")
model = M.DrlClassifier(opt, cfg, seed=0).to(dev)
model.train()
fuse = os.environ.get("FUSE", "1") == "1"
optim = M.FusedAdam(model, lr=opt.vae_lr, fuse_into_backward=fuse)
bs = []
for i in range(4):
    b = synthetic_ecpe_batch(64, 128, cfg.vocab_size, opt.pair_bow_dim, seed=1 + i, shape="A")
    bs.append(({k: v.to(dev) for k, v in b.items()}, b["attention_masks"].sum(1).tolist()))
K = ("input_ids", "attention_masks", "token_type_ids", "emo_labels", "cau_labels", "labels", "bow_reps")
def step(i):
    b, l = bs[i % 4]
    loss = model(*(b[k] for k in K), i % 41, seq_lengths=l)
    optim.zero_grad(); loss.backward(); optim.step()
for i in range(10): step(i)
for trial in range(3):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    host = []
    ev[0].record()
    for i in range(8):
        h0 = time.perf_counter()
        step(i)
        ev[i + 1].record()
        host.append(1e3 * (time.perf_counter() - h0))
    torch.cuda.synchronize()
    print("trial %d  GPU ms: %s | host enqueue ms: %s" % (trial, " ".join("%.2f" % ev[i].elapsed_time(ev[i + 1]) for i in range(8)), " ".join("%.2f" % h for h in host)), flush=True)

# ---- the same through the product's input path (PrefetchLoader over a BatchLoader), as bench.py's headline does
from carel_vae_amd import data as D
for depth in (3, 4):
    ds = D.SyntheticECPEDataset(40 * 64, opt.pair_bow_dim, 1000, vocab_size=cfg.vocab_size, shape="A")
    loader = D.PrefetchLoader(D.BatchLoader(ds, batch_size=64, shuffle=False), dev, depth=depth)
    it = iter(loader)
    def fed(i):
        t0 = time.perf_counter()
        b = next(it)
        t1 = time.perf_counter()
        loss = model(*(b[k] for k in K), i % 41, seq_lengths=b["seq_lengths"])
        optim.zero_grad(); loss.backward(); optim.step()
        return 1e3 * (t1 - t0)
    for i in range(10): fed(i)
    for trial in range(2):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
        host, nxt = [], []
        ev[0].record()
        for i in range(8):
            h0 = time.perf_counter()
            nxt.append(fed(i))
            ev[i + 1].record()
            host.append(1e3 * (time.perf_counter() - h0))
        torch.cuda.synchronize()
        print("fed depth %d trial %d  GPU ms: %s | host step ms: %s | of which next(): %s" % (depth, trial, " ".join("%.2f" % ev[i].elapsed_time(ev[i + 1]) for i in range(8)),
              " ".join("%.2f" % h for h in host), " ".join("%.2f" % h for h in nxt)), flush=True)
    del it, loader
