#!/usr/bin/env python3
"""The fused row-band kernel (gemm_rowln.hip: linear + bias + dropout + residual + LayerNorm, 32 complete rows per workgroup) against the
two-kernel path it replaces (ping-pong GEMM with the residual epilogue, then ln_fwd_kernel): hot (one buffer set) and cold (12 sets used
round robin, as the layers of a step do).  python tools/bench_rowln.py"""
import os, sys, statistics, ctypes as C
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm

lib = L.load()
for v in os.environ.get("VARIANTS", "").split(","):
    if v: L.check(lib.carel_gemm_set_variant(int(v)))
T, NS = int(os.environ.get("T", 8192)), 12
g0 = torch.Generator().manual_seed(0)
def rnd(*s, sc=0.5): return (torch.randn(s, generator=g0) * sc).cuda()
gamma, beta, bias = 1 + rnd(768, sc=0.1), rnd(768, sc=0.1), rnd(768, sc=0.1)
def timed(fns, n=24):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for f in fns: f()
    e0.record()
    for i in range(n): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for K in (768, 3072):
    sets = []
    for i in range(NS):
        sets.append(dict(A=rnd(T, K).bfloat16(), W=rnd(768, K, sc=0.03).bfloat16(), r=rnd(T, 768), h=torch.empty((T, 768), device="cuda"),
                         xf=torch.empty((T, 768), device="cuda"), xb=torch.empty((T, 768), device="cuda", dtype=torch.bfloat16), st=torch.empty((T, 2), device="cuda")))
    def two(s):
        def f():
            gemm(s["A"], s["W"], L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, K, out_f32=s["h"], bias=bias, resid=s["r"], drop=(1, 2, 0, 0.1))
            L.check(lib.carel_layernorm_fwd(s["h"].data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-12, T, 768, s["xf"].data_ptr(), s["xb"].data_ptr(),
                                            s["st"].data_ptr(), L.current_stream()))
        return f
    def fused(s, packed=False):
        a = L.GemmRowLnArgs()
        if packed and "Wp" not in s:
            s["Wp"] = torch.empty(768 * K, device="cuda", dtype=torch.bfloat16)
            L.check(lib.carel_gemm_rowln_pack(s["W"].data_ptr(), K, K, s["Wp"].data_ptr(), L.current_stream()))
        a.w_packed = 1 if packed else 0
        a.A, a.W, a.lda, a.ldb, a.M, a.K = s["A"].data_ptr(), (s["Wp"] if packed else s["W"]).data_ptr(), K, K, T, K
        a.bias, a.resid_f32, a.gamma, a.beta, a.eps = bias.data_ptr(), s["r"].data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-12
        a.h_f32, a.x_f32, a.x_bf16, a.stats = s["h"].data_ptr(), s["xf"].data_ptr(), s["xb"].data_ptr(), s["st"].data_ptr()
        a.drop_seed, a.drop_site, a.drop_idx_offset, a.drop_p, a.drop_row_map = 1, 2, 0, 0.1, None
        def f(): L.check(lib.carel_gemm_rowln(C.byref(a), L.current_stream()))
        return f
    r = {}
    for name, mk in (("gemm + ln", two), ("fused", fused), ("packed", lambda s: fused(s, True))):
        fns = [mk(s) for s in sets]
        hot, cold = [], []
        for _ in range(5):
            hot.append(timed(fns[:1])); cold.append(timed(fns))
        r[name] = (statistics.median(hot), statistics.median(cold))
    fl = 2.0 * T * 768 * K
    print("K = %4d | GEMM + LayerNorm: hot %6.1f us, cold %6.1f us | fused row-band kernel, row-major W: hot %6.1f us, cold %6.1f us | packed W: hot %6.1f us (%4.0f TF), cold %6.1f us (%4.0f TF)" % (
        K, r["gemm + ln"][0], r["gemm + ln"][1], r["fused"][0], r["fused"][1], r["packed"][0], fl / r["packed"][0] / 1e6, r["packed"][1], fl / r["packed"][1] / 1e6), flush=True)
    del sets
    torch.cuda.empty_cache()
