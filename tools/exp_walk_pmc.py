#!/usr/bin/env python3
"""Run each forward/dgrad GEMM shape 3x with the N-fastest walk then 3x with the M-fastest band walk (for rocprofv3 --pmc)."""
import os, sys
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
T = 8192
shapes = [("fwd QKV", L.GEMM_NT, L.EPI_BIAS_BF16, T, 2304, 768), ("fwd FFN1", L.GEMM_NT, L.EPI_BIAS_GELU, T, 3072, 768),
          ("fwd FFN2", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, 3072), ("dgrad FFN2", L.GEMM_NN, L.EPI_DGELU_BF16, T, 3072, 768),
          ("dgrad QKV", L.GEMM_NN, L.EPI_ADD_F32, T, 768, 2304)]
L.check(lib.carel_gemm_set_variant(1))
for name, form, epi, M, N, K in shapes:
    A = rnd(M, K); B = rnd(N, K) if form == L.GEMM_NT else rnd(K, N)
    kw = dict(out_bf16=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16), out2_bf16=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16),
              out_f32=torch.zeros((1, M, N), device="cuda"), bias=torch.zeros(N, device="cuda"), resid=torch.zeros((M, N), device="cuda"),
              aux=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16), drop=(1, 2, 0, 0.1))
    for walk in (20, 24):
        L.check(lib.carel_gemm_set_variant(walk))
        for _ in range(3): gemm(A, B, form, epi, M, N, K, splits=1, **kw)
        torch.cuda.synchronize()
print("order: " + " | ".join("%s x3 N-fastest, x3 M-fastest" % s[0] for s in shapes))
