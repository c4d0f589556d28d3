#!/usr/bin/env python3
"""End-to-end epoch throughput of the reference's driver loop (carel_vae_amd.training.train): a society_num-sized
synthetic ECPE dataset (2 587 pairs, ECPE-shaped lengths, V = 23 771), BERT-base, one MI355X; stock DataLoader vs
carel_vae_amd.BatchLoader vs PrefetchLoader(BatchLoader), each with the fused optimiser.  The test pass after the epoch is part of the loop (:853-914)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch
import carel_vae_amd as cv
from carel_vae_amd import data as D


class SynthDataset(D.ECPEDataset):
    """ECPEDataset with its caches filled from synthetic tensors (no tokenizer / vocabulary on this box)."""

    def __init__(self, n, V, seed):
        b = D.synthetic_ecpe_batch(n, 128, 21128, V, seed=seed, shape="B")
        self.pairs = pd.Series(["x"] * n)
        self.labels = b["labels"].view(-1).numpy(); self.emo_labels = b["emo_labels"].view(-1).numpy(); self.cau_labels = self.labels
        self.max_len, self.bow_features, self.tokenizer = 128, [None] * V, object()
        self.bow_representations = list(b["bow_reps"].numpy())
        self._cache = (b["input_ids"], b["attention_masks"], b["token_type_ids"])


def main():
    if os.environ.get("CAREL_THREADS"): torch.set_num_threads(int(os.environ["CAREL_THREADS"]))
    V, n_train, n_test = 23771, 2587, 1938
    opt = cv.make_opt(epochs=1, pair_bow_dim=V, best_model_path="/tmp/carel_ckpt", model_id="epoch")
    cv.training.save_ckp = lambda *a, **k: None          # the 400 MB checkpoint write (when F1 improves) is not what is measured
    train_ds, test_ds = SynthDataset(n_train, V, 1), SynthDataset(n_test, V, 2)
    names = os.environ.get("CAREL_LOADERS", "DataLoader,BatchLoader,PrefetchLoader").split(",")
    for name in names:
        torch.manual_seed(0)
        model = cv.DrlClassifier(opt, cv.encoder_config("zh"), seed=0).to("cuda")
        optim = cv.FusedAdam(model, lr=opt.vae_lr)
        if name == "DataLoader":
            tr = torch.utils.data.DataLoader(train_ds, batch_size=64, shuffle=True, num_workers=0)
            te = torch.utils.data.DataLoader(test_ds, batch_size=len(test_ds), shuffle=False, num_workers=0)
        elif name == "BatchLoader":
            tr = D.BatchLoader(train_ds, batch_size=64, shuffle=True)
            te = D.BatchLoader(test_ds, batch_size=len(test_ds), shuffle=False)
        else:       # pinned ring + copy stream + sparse bag-of-words around the same BatchLoader
            tr = D.PrefetchLoader(D.BatchLoader(train_ds, batch_size=64, shuffle=True), "cuda", depth=int(os.environ.get("CAREL_PREFETCH_DEPTH", "3")))
            te = D.BatchLoader(test_ds, batch_size=len(test_ds), shuffle=False)
        t0 = time.perf_counter()
        nb = sum(1 for _ in tr)
        torch.cuda.synchronize()
        t_iter = time.perf_counter() - t0
        print("%-14s loader only: %d batches in %.3f s (%.2f ms per batch)" % (name, nb, t_iter, 1e3 * t_iter / nb), flush=True)
        cv.train(tr, te, model, [optim], "cuda", num_unpred_pairs=22, opt=opt, log=lambda *_: None)      # warm-up epoch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cv.train(tr, te, model, [optim], "cuda", num_unpred_pairs=22, opt=opt, log=lambda *_: None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if os.environ.get("CAREL_PROFILE") == name:
            import cProfile, pstats
            pr = cProfile.Profile(); pr.enable()
            cv.train(tr, te, model, [optim], "cuda", num_unpred_pairs=22, opt=opt, log=lambda *_: None)
            torch.cuda.synchronize(); pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
        print("%-14s one epoch (41 steps of 64 + evaluation of %d pairs + checkpoint logic): %.3f s  -> %.0f training pairs/s end to end" % (
            name, n_test, dt, n_train / dt), flush=True)


if __name__ == "__main__":
    main()
