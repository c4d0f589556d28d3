#!/usr/bin/env python3
"""FFN1 -> FFN2 forward chain on per-layer buffers, as the training step runs it (12 layers x {x1 bf16, g, gelu'(u), residual f32, h2 f32}),
each launch bracketed by events: why does FFN2 forward take ~57 us inside the step when its operands are warm (just written) and the
stand-alone launch takes 45-47?  Variants separate the candidates: which of FFN2's buffers are per-layer (cold lines to WRITE /
fresh lines to READ) and which are shared by all layers.  python tools/bench_gemm_chain.py"""
import os, sys, statistics
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm

lib = L.load()
for v in os.environ.get("VARIANTS", "").split(","):
    if v: L.check(lib.carel_gemm_set_variant(int(v)))
T, H, I, NL = 8192, 768, 3072, 12
g0 = torch.Generator().manual_seed(0)
def rnd(*s, sc=0.5): return (torch.randn(s, generator=g0) * sc).cuda()
W1, W2 = rnd(I, H, sc=0.03).bfloat16(), rnd(H, I, sc=0.03).bfloat16()
b1, b2 = torch.zeros(I, device="cuda"), torch.zeros(H, device="cuda")
def bufs(n):
    return dict(x1=[rnd(T, H).bfloat16() for _ in range(n)], g=[torch.empty((T, I), device="cuda", dtype=torch.bfloat16) for _ in range(n)],
                dg=[torch.empty((T, I), device="cuda", dtype=torch.bfloat16) for _ in range(n)], r=[rnd(T, H) for _ in range(n)],
                h2=[torch.empty((T, H), device="cuda") for _ in range(n)])
per, one = bufs(NL), bufs(1)
def run(cfg, reps=6):
    """cfg: for each buffer name, 'per' (one per layer) or 'one' (shared); 'dg' may be None: FFN1 writes g only (inference epilogue)"""
    def pick(name, l): return (per if cfg[name] == "per" else one)[name][l if cfg[name] == "per" else 0]
    t1, t2 = [], []
    for rep in range(reps):
        ev = []
        for l in range(NL):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            if cfg.get("dg") is None:
                gemm(pick("x1", l), W1, L.GEMM_NT, L.EPI_BIAS_GELU, T, I, H, out2_bf16=pick("g", l), bias=b1)
            else:
                gemm(pick("x1", l), W1, L.GEMM_NT, L.EPI_BIAS_GELU_DG, T, I, H, out_bf16=pick("dg", l), out2_bf16=pick("g", l), bias=b1)
            e[1].record()
            gemm(pick("g", l), W2, L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, H, I, out_f32=pick("h2", l), bias=b2, resid=pick("r", l), drop=(1, 2, 0, 0.1))
            e[2].record()
            ev.append(e)
        torch.cuda.synchronize()
        if rep:
            t1 += [e[0].elapsed_time(e[1]) * 1e3 for e in ev]; t2 += [e[1].elapsed_time(e[2]) * 1e3 for e in ev]
    return statistics.median(t1), statistics.median(t2)
base = dict(x1="per", g="per", dg="per", r="per", h2="per")
cases = [("all per-layer (the step)", base), ("all shared (hot)", {k: "one" for k in base}),
         ("h2 shared (FFN2 writes warm lines)", dict(base, h2="one")), ("residual shared", dict(base, r="one")),
         ("g shared", dict(base, g="one")), ("dg shared", dict(base, dg="one")), ("FFN1 writes g only", dict(base, dg=None)),
         ("g + dg shared", dict(base, g="one", dg="one")), ("x1 shared", dict(base, x1="one")),
         ("h2 + residual shared", dict(base, h2="one", r="one"))]
if os.environ.get("QUICK"): cases = cases[:2] + [cases[6]]
for name, cfg in cases:
    a, b = run(cfg)
    print("%-40s FFN1 fwd %6.1f us | FFN2 fwd %6.1f us" % (name, a, b), flush=True)
