#!/usr/bin/env python3
"""Config 4 (drl_classifier_en.py step): fused optimisers with and without the per-layer Adam inside backward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import drl_classifier_en as ME
from carel_vae_amd.data import synthetic_ecpe_batch
V = 22463
for fuse in (True, False, True, False):
    opt = ME.make_opt(pair_bow_dim=V)
    model = ME.DrlClassifier(opt, seed=0).to("cuda"); model.train()
    opts = list(model.make_fused_optimizers(fuse_into_backward=fuse))
    b = {k: v.cuda() for k, v in synthetic_ecpe_batch(64, 128, 50265, V, seed=3, shape="A", pad_id=1, first_id=2, binary_emotion=True).items()}
    def step(i):
        out = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], i % 41)
        cd_e, cd_c, ed, ecd, cad, ced, vae = out
        opts[0].zero_grad(); (cd_e + cd_c).backward(retain_graph=True)
        opts[1].zero_grad(); ed.backward(retain_graph=True)
        opts[3].zero_grad(); ecd.backward(retain_graph=True)
        opts[2].zero_grad(); cad.backward(retain_graph=True)
        opts[4].zero_grad(); ced.backward(retain_graph=True)
        opts[5].zero_grad(); vae.backward()
        for o in opts: o.step()
    for i in range(5): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): step(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("fuse_into_backward=%s: %.3f ms/step" % (fuse, 1e3 * dt), flush=True)
    del model, opts; torch.cuda.empty_cache()
