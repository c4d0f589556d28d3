#!/usr/bin/env python3
"""How long does the HOST take to enqueue one training step (bench configuration), against the GPU's time for it?
If the two are close the step is host-bound wherever the GPU drains its queue (forward -> backward hand-over)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import drl_classifier as M, data as D
dev = "cuda"
opt = M.make_opt(pair_bow_dim=23771)
cfg = M.encoder_config("zh")
torch.manual_seed(0)
model = M.DrlClassifier(opt, cfg, seed=1).to(dev); model.train()
optim = M.FusedAdam(model, lr=1e-5)
SHAPE = os.environ.get("SHAPE", "A")          # A dense, B ECPE-shaped lengths (padding skipped)
b = {k: v.to(dev) for k, v in D.synthetic_ecpe_batch(64, 128, cfg.vocab_size, opt.pair_bow_dim, seed=5, shape=SHAPE).items()}
lengths = b["attention_masks"].sum(1).tolist()
def step(i):
    loss = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], i % 41, seq_lengths=lengths)
    optim.zero_grad(); loss.backward(); optim.step()
for i in range(10): step(i)
torch.cuda.synchronize()
for mode in ("overlap", "serial"):
    model.overlap_wgrad = mode == "overlap"
    for i in range(5): step(i)
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    tf = tb = 0.0
    for i in range(N):
        a0 = time.perf_counter()
        loss = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], i % 41, seq_lengths=lengths)
        a1 = time.perf_counter()
        optim.zero_grad(); loss.backward(); optim.step()
        a2 = time.perf_counter()
        tf += a1 - a0; tb += a2 - a1
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-8s host enqueue %.2f ms/step (forward %.2f, backward+Adam %.2f); GPU drained %.2f ms after the last enqueue; total %.2f ms/step" % (
        mode, 1e3 * (t1 - t0) / N, 1e3 * tf / N, 1e3 * tb / N, 1e3 * (t2 - t1), 1e3 * (t2 - t0) / N))

if os.environ.get("PROFILE"):
    import cProfile, pstats
    model.overlap_wgrad = True
    pr = cProfile.Profile()
    pr.enable()
    for i in range(20): step(i)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
