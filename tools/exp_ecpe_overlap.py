#!/usr/bin/env python3
"""Packed ECPE-shaped step (bench shape B): what does the side stream hide?  ms per step with / without the optimiser, weight gradients on the
side stream or on the main stream, Adam inside backward or in step().  (Round 4: serial and overlapped steps measure the same.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import drl_classifier as M, data as D
dev = "cuda"
opt = M.make_opt(pair_bow_dim=23771)
cfg = M.encoder_config("zh")
torch.manual_seed(0)
SHAPE = os.environ.get("SHAPE", "B")
bs = [{k: v.to(dev) for k, v in D.synthetic_ecpe_batch(64, 128, cfg.vocab_size, opt.pair_bow_dim, seed=101 + i, shape=SHAPE).items()} for i in range(8)]
lens = [b["attention_masks"].sum(1).tolist() for b in bs]
def run(fused, overlap, adam, n=40):
    model = M.DrlClassifier(opt, cfg, seed=1).to(dev); model.train()
    model.overlap_wgrad = overlap
    optim = M.FusedAdam(model, lr=1e-5, fuse_into_backward=fused) if adam else None
    def step(i):
        b = bs[i % 8]
        loss = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], i % 41, seq_lengths=lens[i % 8])
        if optim: optim.zero_grad()
        else: model.zero_grad()
        loss.backward()
        if optim: optim.step()
    for i in range(10): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(i)
    torch.cuda.synchronize()
    del model, optim
    return 1e3 * (time.perf_counter() - t0) / n
cases = [("Adam in backward, side stream", True, True, True), ("Adam in step(), side stream", False, True, True), ("Adam in step(), serial", False, False, True),
         ("no optimiser, side stream", False, True, False), ("no optimiser, serial", False, False, False)]
for r in range(2):
    for name, f, o, a in cases:
        print("%-32s %.3f ms/step" % (name, run(f, o, a)), flush=True)
