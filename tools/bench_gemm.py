#!/usr/bin/env python3
"""Micro-benchmark of the encoder GEMM shapes (T = 8192) for both tile variants; random data."""
import os, sys
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm

lib = L.load()
T = 8192
shapes = [("fwd QKV   NT", L.GEMM_NT, L.EPI_BIAS_BF16, T, 2304, 768, 1),
          ("fwd out   NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, 768, 1),
          ("fwd FFN1  NT", L.GEMM_NT, L.EPI_BIAS_GELU, T, 3072, 768, 1),
          ("fwd FFN2  NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, 3072, 1),
          ("dgrad FFN2 NN", L.GEMM_NN, L.EPI_DGELU_BF16, T, 3072, 768, 1),
          ("dgrad FFN1 NN", L.GEMM_NN, L.EPI_ADD_F32, T, 768, 3072, 1),
          ("dgrad out  NN", L.GEMM_NN, L.EPI_BIAS_BF16, T, 768, 768, 1),
          ("dgrad QKV  NN", L.GEMM_NN, L.EPI_ADD_F32, T, 768, 2304, 1),
          ("wgrad FFN2 TN", L.GEMM_TN, L.EPI_SLAB_F32, 768, 3072, T, 4),
          ("wgrad FFN1 TN", L.GEMM_TN, L.EPI_SLAB_F32, 3072, 768, T, 4),
          ("wgrad out  TN", L.GEMM_TN, L.EPI_SLAB_F32, 768, 768, T, 8),
          ("wgrad QKV  TN", L.GEMM_TN, L.EPI_SLAB_F32, 2304, 768, T, 4)]
splits_try = {L.GEMM_TN: [4, 8, 16]}
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
for name, form, epi, M, N, K, sp in shapes:
    if form == L.GEMM_NT: A, B = rnd(M, K), rnd(N, K)
    elif form == L.GEMM_NN: A, B = rnd(M, K), rnd(K, N)
    else: A, B = rnd(K, M), rnd(K, N)
    res = []
    for variant in (1, 2):
        for s in (splits_try.get(form, [1])):
            kw = dict(out_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16), out2_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16),
                      out_f32=torch.empty((s, M, N), device="cuda"), bias=torch.zeros(N, device="cuda"), resid=torch.zeros((M, N), device="cuda"),
                      aux=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16), drop=(1, 2, 0, 0.1))
            if K % (64 * s): continue
            if variant == 2 and (M % 256 or N % 192): continue
            L.check(lib.carel_gemm_set_variant(variant))
            for _ in range(3): gemm(A, B, form, epi, M, N, K, splits=s, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): gemm(A, B, form, epi, M, N, K, splits=s, **kw)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            res.append("%s%s %6.1f us %5.0f TF" % ("v1" if variant == 1 else "big", ("/s%d" % s) if form == L.GEMM_TN else "", us, 2.0 * M * N * K / us / 1e6))
    print("%-14s M=%5d N=%5d K=%5d | %s" % (name, M, N, K, " | ".join(res)))
L.check(lib.carel_gemm_set_variant(0)); L.check(lib.carel_gemm_set_variant(20))
