#!/usr/bin/env python3
"""Latency floor of the GEMM kernel: tiny grids, several K; and an empty-kernel launch cadence for reference."""
import os, sys
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
L.check(lib.carel_gemm_set_variant(1))
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
x = torch.zeros(64, device="cuda")
print("tiny torch kernel cadence: %.2f us" % timeit(lambda: x.add_(1.0), 200))
for (M, N) in ((128, 128), (2048, 2048), (4096, 1024), (8192, 1024)):
    for K in (64, 768, 3072):
        A, B = rnd(M, K), rnd(N, K)
        kw = dict(out_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16), bias=torch.zeros(N, device="cuda"))
        us = timeit(lambda: gemm(A, B, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, splits=1, **kw))
        print("M=%5d N=%5d K=%5d tiles=%4d  %7.2f us" % (M, N, K, (M // 128) * (N // 128), us))
L.check(lib.carel_gemm_set_variant(0))
