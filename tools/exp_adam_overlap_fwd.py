#!/usr/bin/env python3
"""Timing experiment (numerically RACY on purpose -- never a product path): how much of the fused Adam pass (0.6 ms of pure HBM streaming)
would hide behind the NEXT step's forward pass if it ran on a side stream?  The side stream waits for backward, the next forward does
not wait for the side stream.  If the step gets ~0.5 ms shorter the idea (per-layer Adam chunks in forward order + per-layer events in
carel_encoder_forward) is worth building; if the forward kernels slow down by what Adam's 3 GB do to the Infinity Cache, it is not."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import drl_classifier as M
from carel_vae_amd.data import synthetic_ecpe_batch
dev = torch.device("cuda", 0)
opt, cfg = M.make_opt(), M.encoder_config("zh")
model = M.DrlClassifier(opt, cfg, seed=0).to(dev); model.train()
optim = M.FusedAdam(model, lr=opt.vae_lr)
shape = os.environ.get("SHAPE", "A")
bb, ll = [], []
for i in range(4):
    b = synthetic_ecpe_batch(64, 128, cfg.vocab_size, opt.pair_bow_dim, seed=1 + i, shape=shape)
    ll.append(b["attention_masks"].sum(1).tolist()); bb.append({k: v.to(dev) for k, v in b.items()})
from carel_vae_amd import _lib as L
which = os.environ.get("STREAM", "aux")
side = torch.cuda.Stream(priority=0) if which == "torch" else torch.cuda.ExternalStream(L.load().carel_side_stream(1 if which == "aux" else 0), device=dev)
def run(overlap, steps=40):
    def step(i):
        b = bb[i % 4]
        loss = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], i % 41, seq_lengths=ll[i % 4])
        optim.zero_grad(); loss.backward()
        if overlap:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                optim.step()
        else:
            optim.step()
    for i in range(5): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rep in range(3):
    print("shape %s: plain %.3f ms/step | Adam on a side stream under the next forward (racy) %.3f ms/step" % (shape, run(False), run(True)), flush=True)
