#!/usr/bin/env python3
"""In-kernel time stamps (s_memtime, shader clocks) of ONE steady-state K tile of the ping-pong GEMM, workgroup 0, one wave of each
ping-pong group (needs a -DCAREL_GEMM_ABLATE build: DBG 9).  Per phase: fragment-read issue, DMA issue, counted vmcnt wait,
lgkmcnt wait, barrier, MFMA issue, barrier."""
import os, sys
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
L.check(lib.carel_gemm_set_variant(90 + int(os.environ.get("WIDE", 1))))
for (M, N, K) in [(8192, 768, 3072), (8192, 3072, 768), (8192, 2304, 768)]:
    A, B = rnd(M, K), rnd(N, K)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16); bias = torch.zeros(N, device="cuda")
    ws = torch.zeros(8192, dtype=torch.int64, device="cuda")
    L.check(lib.carel_gemm_set_variant(69))
    for _ in range(3): gemm(A, B, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, out_bf16=out, bias=bias, splitk_ws=ws)
    torch.cuda.synchronize()
    t = ws.cpu().tolist()
    import numpy as np
    nwg = (M // 256) * (N // (96 * (3 if N == 2304 else 2 if N == 3072 else 1)))
    rt = np.array(t[512:512 + nwg * 5], dtype=np.float64).reshape(nwg, 5) * 0.01          # 100 MHz -> us
    t0 = rt[:, 0].min()
    print("  workgroup life (us, %d workgroups; 100-MHz stamps of wave 0): start spread %.2f | prologue (entry -> loop) median %.2f | main loop median %.2f (min %.2f max %.2f) | "
          "epilogue issue median %.2f | store drain median %.2f | first start -> last end %.2f" % (
              nwg, rt[:, 0].max() - t0, np.median(rt[:, 1] - rt[:, 0]), np.median(rt[:, 2] - rt[:, 1]), (rt[:, 2] - rt[:, 1]).min(), (rt[:, 2] - rt[:, 1]).max(),
              np.median(rt[:, 3] - rt[:, 2]), np.median(rt[:, 4] - rt[:, 3]), rt[:, 4].max() - t0))
    if nwg > 256:
        second = rt[:, 0] > t0 + 5.0
        print("  second-round workgroups: %d, their start median %.2f us after the first start" % (int(second.sum()), float(np.median(rt[second, 0]) - t0)))
    names = ["reads issued", "DMA issued", "vmcnt wait", "lgkm wait", "barrier", "MFMAs issued", "barrier"]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gemm(A, B, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, out_bf16=out, bias=bias, splitk_ws=ws)
    e1.record(); torch.cuda.synchronize()
    print("M=%d N=%d K=%d: kernel %.1f us; main loop of workgroup 0 / 100: %d / %d ticks over %d K tiles = %.0f / %.0f per K tile" % (
        M, N, K, e0.elapsed_time(e1) * 100, t[201] - t[200], t[205] - t[204], t[202], (t[201] - t[200]) / max(t[202], 1), (t[205] - t[204]) / max(t[202], 1)))
    for grp in range(2):
        p = 0
        while any(t[(grp * 6 + p) * 8: (grp * 6 + p) * 8 + 8]) and p < 6:
            ts = t[(grp * 6 + p) * 8: (grp * 6 + p) * 8 + 8]
            print("  group %d phase %d: " % (grp, p) + " | ".join("%s %d" % (n, ts[i + 1] - ts[i]) for i, n in enumerate(names)) + " | total %d" % (ts[7] - ts[0]))
            p += 1
L.check(lib.carel_gemm_set_variant(0)); L.check(lib.carel_gemm_set_variant(91))
