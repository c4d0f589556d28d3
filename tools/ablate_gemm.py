#!/usr/bin/env python3
"""Timing ablations of the forward GEMM kernel (interleaved rounds in one process, random data)."""
import os, sys
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
for (M, N, K) in [(8192, 2304, 768), (8192, 768, 3072), (8192, 3072, 768), (8192, 768, 768)]:
    A, B = rnd(M, K), rnd(N, K)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    res = {}
    for rnd_i in range(3):
        for v, name in ((1, "full"), (19, "L2-hot-loads"), (12, "no-global-loads"), (13, "no-mfma"), (11, "no-epilogue")):
            L.check(lib.carel_gemm_set_variant(v))
            for _ in range(2): gemm(A, B, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, out_bf16=out, bias=bias)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): gemm(A, B, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, out_bf16=out, bias=bias)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3 / 20)
    print("M=%d N=%d K=%d: " % (M, N, K) + " | ".join("%s %.1f us (%.0f TF)" % (k, min(v), 2.0 * M * N * K / min(v) / 1e6) for k, v in res.items()))
L.check(lib.carel_gemm_set_variant(0))
