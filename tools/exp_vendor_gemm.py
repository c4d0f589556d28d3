#!/usr/bin/env python3
"""Reference point: what the vendor BLAS behind torch (hipBLASLt / rocBLAS) reaches on the encoder's GEMM shapes, plain
bf16 GEMM without any epilogue, vs carel_gemm_bf16 with its fused epilogue.  Measurement only -- the product path never
calls torch.matmul."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
T = 8192
shapes = [("fwd QKV   NT", "NT", L.GEMM_NT, L.EPI_BIAS_BF16, T, 2304, 768), ("fwd out   NT", "NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, 768),
          ("fwd FFN1  NT", "NT", L.GEMM_NT, L.EPI_BIAS_GELU, T, 3072, 768), ("fwd FFN2  NT", "NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, 3072),
          ("dgrad FFN2 NN", "NN", L.GEMM_NN, L.EPI_DGELU_BF16, T, 3072, 768), ("dgrad FFN1 NN", "NN", L.GEMM_NN, L.EPI_ADD_F32, T, 768, 3072),
          ("dgrad QKV  NN", "NN", L.GEMM_NN, L.EPI_ADD_F32, T, 768, 2304),
          ("wgrad FFN  TN", "TN", L.GEMM_TN, L.EPI_SLAB_F32, 768, 3072, T), ("wgrad QKV  TN", "TN", L.GEMM_TN, L.EPI_SLAB_F32, 2304, 768, T),
          ("wgrad out  TN", "TN", L.GEMM_TN, L.EPI_SLAB_F32, 768, 768, T)]
for name, kind, form, epi, M, N, K in shapes:
    if kind == "NT": A, B = rnd(M, K), rnd(N, K); f = lambda: torch.mm(A, B.t())
    elif kind == "NN": A, B = rnd(M, K), rnd(K, N); f = lambda: torch.mm(A, B)
    else: A, B = rnd(K, M), rnd(K, N); f = lambda: torch.mm(A.t(), B)
    tv = timeit(f)
    sp = 4 if kind == "TN" else 1
    kw = dict(out_bf16=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16), out2_bf16=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16),
              out_f32=torch.zeros((sp, M, N), device="cuda"), bias=torch.zeros(N, device="cuda"), resid=torch.zeros((M, N), device="cuda"),
              aux=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16), drop=(1, 2, 0, 0.1))
    tc = timeit(lambda: gemm(A, B, form, epi, M, N, K, splits=sp, **kw))
    fl = 2.0 * M * N * K
    print("%-14s M=%5d N=%5d K=%5d | torch.mm (vendor BLAS, no epilogue) %6.1f us %5.0f TF | carel (fused epilogue%s) %6.1f us %5.0f TF" % (
        name, M, N, K, tv, fl / tv / 1e6, ", 4 slabs, w/o reduce" if kind == "TN" else "", tc, fl / tc / 1e6))
