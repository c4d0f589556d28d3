#!/usr/bin/env python3
"""What does the dense step lose when a few CUs are held by somebody else for part of the backward pass -- the situation an overlapped gradient
all-reduce creates (its kernel keeps one persistent workgroup per channel resident for the length of the collective)?  `n` idle workgroups
(tools/ubench/cu_hog.hip: 256 threads, optional LDS) are launched on their own stream right before loss.backward() and stay for `usec`.
The step's GEMMs are ONE round of 256 tiles on 256 CUs with the whole LDS and register file of a CU each.  ms per step, median of 3 x 20 steps."""
import ctypes as C, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import drl_classifier as M, data as D
hog = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libcu_hog.so"))
hog.cu_hog_launch.argtypes = [C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
dev = "cuda"
opt = M.make_opt(pair_bow_dim=23771)
cfg = M.encoder_config("zh")
torch.manual_seed(0)
model = M.DrlClassifier(opt, cfg, seed=1).to(dev); model.train()
DP_LIKE = os.environ.get("DP_LIKE") == "1"          # the schedule DataParallel runs: weight gradients on the main stream, Adam in step()
model.overlap_wgrad = not DP_LIKE
optim = M.FusedAdam(model, lr=1e-5, fuse_into_backward=not DP_LIKE)
bs = [{k: v.to(dev) for k, v in D.synthetic_ecpe_batch(64, 128, cfg.vocab_size, opt.pair_bow_dim, seed=5 + i, shape="A").items()} for i in range(4)]
side = torch.cuda.Stream()
sink = torch.zeros(4, device=dev, dtype=torch.int32)
def step(i, n, usec, lds):
    b = bs[i % 4]
    loss = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], i % 41)
    optim.zero_grad()
    if n:
        side.wait_stream(torch.cuda.current_stream())          # the collective starts when the first gradients exist: here, with the backward pass
        rc = hog.cu_hog_launch(n, usec, lds, sink.data_ptr(), side.cuda_stream)
        assert rc == 0, rc
    loss.backward(); optim.step()
def run(n, usec, lds, steps=20):
    for i in range(5): step(i, n, usec, lds)
    torch.cuda.synchronize()
    r = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(steps): step(i, n, usec, lds)
        torch.cuda.synchronize()
        r.append(1e3 * (time.perf_counter() - t0) / steps)
    return statistics.median(r)
print("schedule:", "DataParallel-like (weight gradients on the main stream, Adam in step())" if DP_LIKE else "bench default (side stream, Adam inside backward)")
base = run(0, 0, 0)
print("no hog                                   %.3f ms" % base, flush=True)
for usec in (1000.0, 2500.0):
    for lds in (0, 32768):
        for n in (1, 4, 8, 16, 32, 64):
            t = run(n, usec, lds)
            print("%2d workgroups x %4.1f ms, LDS %2d KB:        %.3f ms  (+%.2f)" % (n, usec / 1e3, lds // 1024, t, t - base), flush=True)
