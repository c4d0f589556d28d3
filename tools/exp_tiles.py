#!/usr/bin/env python3
"""Tile-count experiment: time of the 128x128 GEMM kernel vs number of output tiles (K = 768 and 3072), plain bf16 output."""
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
for K in (768, 3072):
    for (M, N) in ((2048, 1024), (4096, 1024), (6144, 1024), (8192, 1024), (8192, 768), (8192, 1536), (8192, 2304), (8192, 3072)):
        A, B = rnd(M, K), rnd(N, K)
        kw = dict(out_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16), bias=torch.zeros(N, device="cuda"))
        for _ in range(3): gemm(A, B, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): gemm(A, B, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        tiles = (M // 128) * (N // 128)
        print("K=%4d M=%5d N=%5d tiles=%5d (%.2f x 512) %7.1f us %6.0f TF  per-CU operand bytes/us at 2 tiles: %.0f GB/s/CU" % (
            K, M, N, tiles, tiles / 512, us, 2.0 * M * N * K / us / 1e6, tiles * 256 * K * 2 / 256 / us / 1e3))
L.check(lib.carel_gemm_set_variant(0))
