#!/usr/bin/env python3
"""Encoder GEMM shapes at T = 8192 with HOT operands (one buffer set re-used by every launch: everything sits in the 256-MiB Infinity
Cache, which is what tools/bench_gemm_pp.py measures) against COLD operands (NSETS distinct sets of activations / outputs used round
robin, as the layers of a training step do; the weights rotate too).  Answers: how much of the in-step GEMM time (tools/trace_step_seq.py)
is the memory system rather than the main loop.  NSETS=12 python tools/bench_gemm_cold.py"""
import os, sys, statistics
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm

lib = L.load()
for v in os.environ.get("VARIANTS", "").split(","):
    if v: L.check(lib.carel_gemm_set_variant(int(v)))       # e.g. VARIANTS=170: epilogue inputs requested after the main loop (round 2)
T = int(os.environ.get("T", 8192))
NSETS = int(os.environ.get("NSETS", 12))
shapes = [("fwd QKV   NT", L.GEMM_NT, L.EPI_BIAS_BF16, T, 2304, 768),
          ("fwd out   NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, 768),
          ("fwd FFN1+dg NT", L.GEMM_NT, L.EPI_BIAS_GELU_DG, T, 3072, 768),
          ("fwd FFN2  NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, 3072),
          ("dgrad FFN2* NN", L.GEMM_NN, L.EPI_MUL_BF16, T, 3072, 768),
          ("dgrad FFN1 NN", L.GEMM_NN, L.EPI_ADD_F32, T, 768, 3072),
          ("dgrad out  NN", L.GEMM_NN, L.EPI_BIAS_BF16, T, 768, 768),
          ("dgrad QKV  NN", L.GEMM_NN, L.EPI_ADD_F32, T, 768, 2304)]
if os.environ.get("SHAPES"):
    keep = [int(x) for x in os.environ["SHAPES"].split(",")]
    shapes = [shapes[i] for i in keep]
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
def timed(fns, n=24):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for f in fns: f()
    e0.record()
    for i in range(n): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tot = {"hot": 0.0, "cold": 0.0}
for name, form, epi, M, N, K in shapes:
    sets = []
    ws = torch.zeros(96 << 20, dtype=torch.uint8, device="cuda") if os.environ.get("PAIR", "0") == "1" else None      # PAIR=1 (with VARIANTS=201): zero-filled workspace + pair split-K
    for i in range(NSETS):
        A = rnd(M, K)
        B = rnd(N, K) if form == L.GEMM_NT else rnd(K, N)
        kw = dict(out_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16), out2_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16),
                  out_f32=torch.empty((M, N), device="cuda"), bias=torch.zeros(N, device="cuda"), resid=torch.zeros((M, N), device="cuda"),
                  aux=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16), drop=(1, 2, 0, 0.1))
        if ws is not None and epi in (L.EPI_BIAS_DROP_RESID, L.EPI_ADD_F32): kw.update(splitk_ws=ws, ws_zeroed=True)
        sets.append((A, B, kw))
    def mk(s):
        A, B, kw = s
        return lambda: gemm(A, B, form, epi, M, N, K, **kw)
    fns = [mk(s) for s in sets]
    # which operands rotate: all of them, or only A / only B / only the epilogue's inputs and outputs (the rest stay set 0)
    def mk_part(s, part):
        A, B, kw = sets[0]
        if part == "A": A = s[0]
        if part == "B": B = s[1]
        if part == "E": kw = s[2]
        if part == "I": kw = dict(kw, resid=s[2]["resid"], aux=s[2]["aux"])                     # only the epilogue's INPUTS rotate
        if part == "O": kw = dict(kw, out_bf16=s[2]["out_bf16"], out2_bf16=s[2]["out2_bf16"], out_f32=s[2]["out_f32"])   # only its OUTPUTS
        return lambda: gemm(A, B, form, epi, M, N, K, **kw)
    hot, cold = [], []
    parts = {k: [] for k in "ABEIO"}
    for r in range(5):
        hot.append(timed(fns[:1])); cold.append(timed(fns))
        for k in parts: parts[k].append(timed([mk_part(s, k) for s in sets]))
    h, c = statistics.median(hot), statistics.median(cold)
    extra = " | only A cold +%.1f, only B +%.1f, only epilogue buffers +%.1f (inputs +%.1f, outputs +%.1f)" % tuple(statistics.median(parts[k]) - h for k in "ABEIO")
    tot["hot"] += h; tot["cold"] += c
    fl = 2.0 * M * N * K
    print("%-14s M=%5d N=%5d K=%5d | hot %6.1f us %5.0f TF | cold (%d sets) %6.1f us %5.0f TF | +%.1f us" % (name, M, N, K, h, fl / h / 1e6, NSETS, c, fl / c / 1e6, c - h) + extra, flush=True)
    del sets, fns
    torch.cuda.empty_cache()
print("sum: hot %.1f us  cold %.1f us" % (tot["hot"], tot["cold"]))
