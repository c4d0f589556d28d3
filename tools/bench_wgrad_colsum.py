#!/usr/bin/env python3
"""QKV weight gradient (2304 x 768, K = 8192 tokens) with and without the bias gradient from the same GEMM (ones-vector MFMA in the
first tile column).  Measurement only."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
T = 8192
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for M, N in [(2304, 768), (768, 768)]:
    A, B = rnd(T, M), rnd(T, N)
    sp = lib.carel_gemm_wgrad_splits(M, N, T)
    slabs = torch.empty((sp, M, N), device="cuda"); cs = torch.empty((sp, M), device="cuda")
    f0 = lambda: gemm(A, B, L.GEMM_TN, L.EPI_SLAB_F32, M, N, T, splits=sp, out_f32=slabs)
    f1 = lambda: gemm(A, B, L.GEMM_TN, L.EPI_SLAB_F32, M, N, T, splits=sp, out_f32=slabs, colsum_a=cs)
    r0, r1 = [], []
    for _ in range(5): r0.append(timed(f0)); r1.append(timed(f1))
    print("dW %d x %d, %d slices: GEMM alone %.1f us, with the bias-gradient sums %.1f us" % (M, N, sp, statistics.median(r0), statistics.median(r1)))
