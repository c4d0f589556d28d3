#!/usr/bin/env python3
"""Timing ablations (wrong results; ablation build) of the three-group GEMM kernel on the FFN2-forward shape, next to the ping-pong kernel."""
import sys, os, statistics, torch
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
g = torch.Generator().manual_seed(0)
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
names = {220: "ping-pong", 221: "tri full", 222: "tri no DMA", 223: "tri no MFMA", 224: "tri no reads", 225: "tri barriers only"}
for (M, N, K) in [(8192, 768, 3072), (8192, 768, 768)]:
    A = (torch.randn((M, K), generator=g) * 0.5).cuda().bfloat16(); B = (torch.randn((N, K), generator=g) * 0.5).cuda().bfloat16()
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16); bias = torch.zeros(N, device="cuda")
    res = {v: [] for v in names}
    for r in range(4):
        for v in names:
            L.check(lib.carel_gemm_set_variant(v))
            f = lambda: gemm(A, B, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, out_bf16=out, bias=bias)
            f(); t = timed(f)
            if r: res[v].append(t)
    print("M=%d N=%d K=%d: " % (M, N, K) + " | ".join("%s %.1f" % (names[v], statistics.median(res[v])) for v in names), flush=True)
L.check(lib.carel_gemm_set_variant(220))
