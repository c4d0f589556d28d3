#!/usr/bin/env python3
"""What each fused epilogue of the ping-pong kernel costs on top of the plain bias -> bf16 GEMM of the same shape (T = 8192).  Measurement only."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
M = 8192
g = torch.Generator().manual_seed(0)
def rnd(*s, sc=0.5): return (torch.randn(s, generator=g) * sc).cuda().bfloat16()
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for form, N, K in [("NT", 768, 768), ("NT", 768, 3072), ("NN", 768, 3072), ("NN", 768, 2304), ("NN", 3072, 768)]:
    A = rnd(M, K); B = rnd(N, K, sc=0.05) if form == "NT" else rnd(K, N, sc=0.05)
    F = L.GEMM_NT if form == "NT" else L.GEMM_NN
    ob = torch.empty((M, N), device="cuda", dtype=torch.bfloat16); of = torch.empty((M, N), device="cuda")
    r = torch.zeros((M, N), device="cuda"); bias = torch.zeros(N, device="cuda"); aux = rnd(M, N); cs = torch.empty((M // 128, N), device="cuda")
    cases = [("bias -> bf16", lambda: gemm(A, B, F, L.EPI_BIAS_BF16, M, N, K, out_bf16=ob, bias=bias))]
    if form == "NT":
        cases += [("bias + residual -> f32, dropout p = 0", lambda: gemm(A, B, F, L.EPI_BIAS_DROP_RESID, M, N, K, out_f32=of, bias=bias, resid=r, drop=(1, 2, 0, 0.0))),
                  ("bias + dropout 0.1 + residual -> f32", lambda: gemm(A, B, F, L.EPI_BIAS_DROP_RESID, M, N, K, out_f32=of, bias=bias, resid=r, drop=(1, 2, 0, 0.1)))]
    else:
        cases += [("-> f32 (no residual)", lambda: gemm(A, B, F, L.EPI_ADD_F32, M, N, K, out_f32=of)),
                  ("+ f32 residual -> f32", lambda: gemm(A, B, F, L.EPI_ADD_F32, M, N, K, out_f32=of, resid=r))]
        if N % 192 == 0:
            cases += [("x saved gelu' (bf16) -> bf16 + column sums", lambda: gemm(A, B, F, L.EPI_MUL_BF16, M, N, K, out_bf16=ob, aux=aux, colsum_part=cs)),
                      ("x saved gelu' (bf16) -> bf16, no column sums", lambda: gemm(A, B, F, L.EPI_MUL_BF16, M, N, K, out_bf16=ob, aux=aux))]
    print("%s 8192 x %d x %d" % (form, N, K))
    res = {n: [] for n, _ in cases}
    for rep in range(3):
        for n, fn in cases: res[n].append(timed(fn))
    for n, _ in cases: print("   %-48s %6.1f us" % (n, statistics.median(res[n])), flush=True)
