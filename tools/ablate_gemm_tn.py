#!/usr/bin/env python3
"""Where the weight-gradient (A^T B, K = tokens) ping-pong GEMM's time goes: timing ablations (wrong results) of the FFN gradient
768 x 3072 x 8192 / 3072 x 768 x 8192 at split-K 5 (240 workgroups of the 256 x 192 tile: the production launch).  Needs a library built with
CAREL_BUILD_TAG=ablate CAREL_EXTRA_FLAGS=-DCAREL_GEMM_ABLATE python -m carel_vae_amd.build, loaded through CAREL_HIP_LIB."""
import os, sys, statistics
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
names = {0: "full", 61: "no DMA", 62: "no MFMA", 63: "no reads", 64: "no epilogue", 66: "DMA+barriers only", 67: "barriers only", 68: "MFMA+barriers only"}
T = 8192
for (M, N) in [(768, 3072), (3072, 768)]:
    A, B = rnd(T, M), rnd(T, N)
    for sp in (5,):
        slabs = torch.empty((sp, M, N), device="cuda")
        res = {v: [] for v in names}
        for r in range(4):
            for v in names:
                L.check(lib.carel_gemm_set_variant(v))
                f = lambda: gemm(A, B, L.GEMM_TN, L.EPI_SLAB_F32, M, N, T, splits=sp, out_f32=slabs)
                f(); t = timed(f)
                if r: res[v].append(t)
        nk = T // 64 / sp
        print("dW %d x %d, split %d (%d workgroups, %.1f K tiles each): " % (M, N, sp, (M // 256) * (N // 192) * sp, nk) +
              " | ".join("%s %.1f" % (names[v], statistics.median(res[v])) for v in names), flush=True)
L.check(lib.carel_gemm_set_variant(0))
