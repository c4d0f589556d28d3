#!/usr/bin/env python3
"""The encoder's forward / data-gradient GEMMs at the row counts of packed ECPE batches (M = 1664 .. 2048): the library's default choice
(128x128 kernel or split-K slabs on the ping-pong kernel + slab epilogue) against the ping-pong kernel taking every grid of >= 32 tiles
directly (hook 51) and the 128x128 kernel forced (variant 1).  Median of interleaved rounds, us per call (both launches where two are made)."""
import os, sys, statistics
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
def timed(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
shapes = [("fwd QKV   NT", L.GEMM_NT, L.EPI_BIAS_BF16, 2304, 768), ("fwd out   NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, 768, 768),
          ("fwd FFN1  NT", L.GEMM_NT, L.EPI_BIAS_GELU_DG, 3072, 768), ("fwd FFN2  NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, 768, 3072),
          ("dgrad FFN2 NN", L.GEMM_NN, L.EPI_MUL_BF16, 3072, 768), ("dgrad FFN1 NN", L.GEMM_NN, L.EPI_ADD_F32, 768, 3072),
          ("dgrad out  NN", L.GEMM_NN, L.EPI_BIAS_BF16, 768, 768), ("dgrad QKV  NN", L.GEMM_NN, L.EPI_ADD_F32, 768, 2304)]
variants = [("default", (0, 53)), ("pp >= 32 tiles", (0, 51)), ("128x128", (1, 53))]
for M in [int(x) for x in os.environ.get("MS", "1664,2048").split(",")]:
    tot = {v[0]: 0.0 for v in variants}
    for name, form, epi, N, K in shapes:
        A = rnd(M, K); B = rnd(N, K) if form == L.GEMM_NT else rnd(K, N)
        ws = torch.zeros(64 << 20, device="cuda", dtype=torch.uint8)
        kw = dict(out_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16), out2_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16),
                  out_f32=torch.empty((M, N), device="cuda"), bias=torch.zeros(N, device="cuda"), resid=torch.zeros((M, N), device="cuda"),
                  aux=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16), drop=(1, 2, 0, 0.1), splitk_ws=ws)
        if epi == L.EPI_MUL_BF16: kw["colsum_part"] = torch.empty(((M + 127) // 128, N), device="cuda")
        res = {v[0]: [] for v in variants}
        for r in range(5):
            for vn, (gv, hook) in variants:
                L.check(lib.carel_gemm_set_variant(gv)); L.check(lib.carel_gemm_set_variant(hook))
                f = lambda: gemm(A, B, form, epi, M, N, K, **kw)
                try:
                    f(); t = timed(f)
                except L.CarelError as e:
                    t = float("nan")
                if r: res[vn].append(t)
        med = {k: statistics.median(v) for k, v in res.items()}
        for k in tot: tot[k] += med[k]
        print("M=%4d %-14s N=%4d K=%4d | " % (M, name, N, K) + " | ".join("%s %6.1f" % (k, v) for k, v in med.items()), flush=True)
    print("M=%4d sum: " % M + " | ".join("%s %6.1f" % (k, v) for k, v in tot.items()), flush=True)
L.check(lib.carel_gemm_set_variant(0)); L.check(lib.carel_gemm_set_variant(53))
