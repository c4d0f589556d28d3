#!/usr/bin/env python3
"""Where the ping-pong GEMM's time goes: timing ablations (wrong results) of the QKV-forward (npn 3) and FFN2-forward-like
(npn 1) launches.  Needs a library built with CAREL_BUILD_TAG=ablate CAREL_EXTRA_FLAGS=-DCAREL_GEMM_ABLATE python -m carel_vae_amd.build and (with -DCAREL_EXPERIMENTS too) loaded with CAREL_HIP_EXP_LIB=carel_vae_amd/libcarel_hip_ablate.so (the product library is not touched)."""
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
names = {3: "full", 61: "no DMA", 62: "no MFMA", 63: "no reads", 64: "no epilogue", 66: "DMA+barriers only", 67: "barriers only", 68: "MFMA+barriers only"}
WIDE = int(os.environ.get("WIDE", 0))
L.check(lib.carel_gemm_set_variant(90 + WIDE))
print("wide-phase schedule" if WIDE else "fine schedule")
MS = [int(x) for x in os.environ.get("MS", "8192").split(",")]          # MS=1664 for the packed ECPE row count
if MS != [8192]: L.check(lib.carel_gemm_set_variant(51))                  # (small grids: the ping-pong kernel takes every grid of >= 32 tiles)
for (M, N, K) in [(m, n, k) for m in MS for (n, k) in [(2304, 768), (3072, 768), (768, 3072), (768, 768)]]:
    A, B = rnd(M, K), rnd(N, K)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    res = {v: [] for v in names}
    for r in range(5):
        for v in names:
            L.check(lib.carel_gemm_set_variant(v))
            f = lambda: gemm(A, B, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, out_bf16=out, bias=bias)
            f(); t = timed(f)
            if r: res[v].append(t)
    print("M=%d N=%d K=%d: " % (M, N, K) + " | ".join("%s %.1f us" % (names[v], statistics.median(res[v])) for v in names), flush=True)
L.check(lib.carel_gemm_set_variant(0)); L.check(lib.carel_gemm_set_variant(91))
