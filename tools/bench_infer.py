#!/usr/bin/env python3
"""pair_probabilities throughput (the kernel path of get_pair_preds, ref :265-282): 2048 ECPE-shaped pairs in chunks of 256."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import drl_classifier as M, data as D
opt = M.make_opt(pair_bow_dim=23771); cfg = M.encoder_config("zh")
model = M.DrlClassifier(opt, cfg, seed=1).to("cuda"); model.eval()
b = D.synthetic_ecpe_batch(2048, 128, cfg.vocab_size, 8, seed=9, shape="B")
ids, att, tt = (b[k].cuda() for k in ("input_ids", "attention_masks", "token_type_ids"))
for chunk in (128, 256, 384, 512, 1024):
    for _ in range(2): model.pair_probabilities(ids, att, tt, chunk=chunk)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); model.pair_probabilities(ids, att, tt, chunk=chunk); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("inference, chunks of %4d pairs: %.2f ms per 2048 pairs (median) = %.0f pairs/s" % (chunk, 1e3 * ts[2], 2048 / ts[2]), flush=True)
