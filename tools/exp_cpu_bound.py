#!/usr/bin/env python3
"""Is the step CPU-bound?  Host time to ENQUEUE n steps vs time until the GPU has finished them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import drl_classifier as M
from oracle import carel_oracle as O
dev = torch.device("cuda", 0)
opt, cfg = M.make_opt(), M.encoder_config("zh")
model = M.DrlClassifier(opt, cfg, seed=0).to(dev); model.train()
optim = M.FusedAdam(model, lr=opt.vae_lr, fuse_into_backward=True)
b = {k: v.to(dev) for k, v in O.synthetic_batch(64, 128, O.EncoderConfig(), opt.pair_bow_dim, seed=1).items()}
lens = b["attention_masks"].sum(1).tolist()
def step(i):
    loss = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], i % 41, seq_lengths=lens)
    optim.zero_grad(); loss.backward(); optim.step()
for i in range(5): step(i)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.3f ms/step   total %.3f ms/step   (GPU backlog at the end of enqueue: %.3f ms)" % (1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n, 1e3 * (t2 - t1)))
# fwd / bwd split of the enqueue time
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
