#!/usr/bin/env python3
"""Encoder GEMM shapes at T = 8192 (random data): the 128x128 kernel (variant 1) vs the ping-pong kernel (variant 3; fine and wide schedule),
interleaved rounds in one process (median of 5 rounds x 20 launches), plus the vendor BLAS (torch.mm, plain GEMM without
epilogue) as the reference point.  Measurement only; the product never calls the vendor library."""
import os, sys, statistics
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm

lib = L.load()
T = int(os.environ.get("T", 8192))
shapes = [("fwd QKV   NT", L.GEMM_NT, L.EPI_BIAS_BF16, T, 2304, 768),
          ("fwd out   NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, 768),
          ("fwd FFN1  NT", L.GEMM_NT, L.EPI_BIAS_GELU, T, 3072, 768),
          ("fwd FFN1+dg NT", L.GEMM_NT, L.EPI_BIAS_GELU_DG, T, 3072, 768),
          ("fwd FFN2  NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, 3072),
          ("dgrad FFN2 NN", L.GEMM_NN, L.EPI_DGELU_BF16, T, 3072, 768),
          ("dgrad FFN2* NN", L.GEMM_NN, L.EPI_MUL_BF16, T, 3072, 768),
          ("dgrad FFN1 NN", L.GEMM_NN, L.EPI_ADD_F32, T, 768, 3072),
          ("dgrad out  NN", L.GEMM_NN, L.EPI_BIAS_BF16, T, 768, 768),
          ("dgrad QKV  NN", L.GEMM_NN, L.EPI_ADD_F32, T, 768, 2304),
          ("plain QKV  NT", L.GEMM_NT, L.EPI_ADD_F32, T, 2304, 768)]
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tot = {1: 0.0, 3: 0.0, "ppw": 0.0, "blas": 0.0}
for name, form, epi, M, N, K in shapes:
    A = rnd(M, K)
    B = rnd(N, K) if form == L.GEMM_NT else rnd(K, N)
    kw = dict(out_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16), out2_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16),
              out_f32=torch.empty((M, N), device="cuda"), bias=torch.zeros(N, device="cuda"), resid=torch.zeros((M, N), device="cuda"),
              aux=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16), drop=(1, 2, 0, 0.1))
    if epi in (L.EPI_DGELU_BF16, L.EPI_MUL_BF16): kw["colsum_part"] = torch.empty((M // 128, N), device="cuda")
    def run(v):                                     # 3 ("pp"): the ping-pong kernel with row-major XCD chunks (hook 120); "ppw": XCD rectangles (121, default)
        L.check(lib.carel_gemm_set_variant(3 if v == "ppw" else v))
        L.check(lib.carel_gemm_set_variant(120 if v == 3 else 121))
        return lambda: gemm(A, B, form, epi, M, N, K, **kw)
    Bm = B.t() if form == L.GEMM_NT else B
    fns = {1: run, 3: run}
    ts = {1: [], 3: [], "ppw": [], "blas": []}
    for rnd_i in range(6):
        for v in (1, 3, "ppw"):
            f = run(v); f(); t = timed(f)
            if rnd_i: ts[v].append(t)
        f = lambda: torch.mm(A, Bm); f(); t = timed(f)
        if rnd_i: ts["blas"].append(t)
    med = {k: statistics.median(v) for k, v in ts.items()}
    fl = 2.0 * M * N * K
    for k in tot: tot[k] += med[k]
    print("%-14s M=%5d N=%5d K=%5d | v1 %6.1f us %5.0f TF | pp/chunks %6.1f us %5.0f TF | pp/rect %6.1f us %5.0f TF (min %6.1f) | blas %6.1f us %5.0f TF" % (
        name, M, N, K, med[1], fl / med[1] / 1e6, med[3], fl / med[3] / 1e6, med["ppw"], fl / med["ppw"] / 1e6, min(ts["ppw"]),
        med["blas"], fl / med["blas"] / 1e6), flush=True)
L.check(lib.carel_gemm_set_variant(121))
print("sum: v1 %.1f us  pp/chunks %.1f us  pp/rect %.1f us  blas %.1f us" % (tot[1], tot[3], tot["ppw"], tot["blas"]))
# tile width experiments: npn forced (variant 70 + n) on the wide GEMMs
for name, form, epi, M, N, K in [s for s in shapes if s[4] >= 2304]:
    A = rnd(M, K); B = rnd(N, K) if form == L.GEMM_NT else rnd(K, N)
    kw = dict(out_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16), out2_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16),
              out_f32=torch.empty((M, N), device="cuda"), bias=torch.zeros(N, device="cuda"), resid=torch.zeros((M, N), device="cuda"),
              aux=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16), drop=(1, 2, 0, 0.1))
    L.check(lib.carel_gemm_set_variant(3))
    r = []
    for n in (1, 2, 3):
        L.check(lib.carel_gemm_set_variant(70 + n))
        f = lambda: gemm(A, B, form, epi, M, N, K, **kw)
        f(); ts = [timed(f) for _ in range(4)]
        r.append("npn%d %6.1f us" % (n, statistics.median(ts)))
    L.check(lib.carel_gemm_set_variant(70))
    print("%-14s %s" % (name, " | ".join(r)), flush=True)
# weight gradients: GEMM into slabs + the slab reduction, split factor chosen by the library for each kernel
wt = {1: 0.0, 3: 0.0, 4: 0.0}       # 4 = ping-pong with the wide-phase schedule
for name, M, N in [("wgrad FFN2 TN", 768, 3072), ("wgrad FFN1 TN", 3072, 768), ("wgrad out  TN", 768, 768), ("wgrad QKV  TN", 2304, 768)]:
    A, B = rnd(T, M), rnd(T, N)
    dW = torch.empty((M, N), device="cuda")
    res = {}
    for rnd_i in range(4):
        for v in (1, 3, 4):
            L.check(lib.carel_gemm_set_variant(3 if v == 4 else v))
            L.check(lib.carel_gemm_set_variant(91 if v == 4 else 90))      # 3: fine schedule, 4: wide (default)
            sp = lib.carel_gemm_wgrad_splits(M, N, T)
            slabs = torch.empty((sp, M, N), device="cuda")
            def f():
                gemm(A, B, L.GEMM_TN, L.EPI_SLAB_F32, M, N, T, splits=sp, out_f32=slabs)
                L.check(lib.carel_slab_reduce_f32(slabs.data_ptr(), dW.data_ptr(), M * N, sp, 0, L.current_stream()))
            f(); t = timed(f)
            if rnd_i: res.setdefault(v, []).append((t, sp))
    m1, m3, m4 = (statistics.median(x[0] for x in res[k]) for k in (1, 3, 4))
    wt[1] += m1; wt[3] += m3; wt[4] += m4
    fl = 2.0 * M * N * T
    print("%-14s M=%5d N=%5d K=%5d | v1/s%d %6.1f us %5.0f TF | pp/s%d %6.1f us %5.0f TF | ppw %6.1f us %5.0f TF" % (name, M, N, T, res[1][0][1], m1, fl / m1 / 1e6, res[3][0][1], m3, fl / m3 / 1e6, m4, fl / m4 / 1e6), flush=True)
print("wgrad sum (GEMM + reduce): v1 %.1f us  pp %.1f us  ppw %.1f us" % (wt[1], wt[3], wt[4]))
L.check(lib.carel_gemm_set_variant(0)); L.check(lib.carel_gemm_set_variant(91))
