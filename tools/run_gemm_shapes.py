#!/usr/bin/env python3
"""Launch every encoder GEMM shape (T = 8192, random data) a few times with the kernel variant given in argv[1]
(default 3 = ping-pong kernel; further tuning hooks in argv[3:], e.g. 120 = row-major XCD chunks): the program to put under
rocprofv3 (tools/pmc_gemm_pp.sh, tools/pmc_gemm_fetch.sh).  The training encoder's epilogues (gelu'(u) saved by FFN1)."""
import os, sys
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm

lib = L.load()
T = 8192
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
shapes = [(L.GEMM_NT, L.EPI_BIAS_BF16, T, 2304, 768), (L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, 768), (L.GEMM_NT, L.EPI_BIAS_GELU_DG, T, 3072, 768),
          (L.GEMM_NT, L.EPI_BIAS_DROP_RESID, T, 768, 3072), (L.GEMM_NN, L.EPI_MUL_BF16, T, 3072, 768), (L.GEMM_NN, L.EPI_ADD_F32, T, 768, 3072),
          (L.GEMM_NN, L.EPI_BIAS_BF16, T, 768, 768), (L.GEMM_NN, L.EPI_ADD_F32, T, 768, 2304)]
L.check(lib.carel_gemm_set_variant(variant))
for hook in sys.argv[3:]: L.check(lib.carel_gemm_set_variant(int(hook)))
for form, epi, M, N, K in shapes:
    A = rnd(M, K)
    B = rnd(N, K) if form == L.GEMM_NT else rnd(K, N)
    kw = dict(out_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16), out2_bf16=torch.empty((M, N), device="cuda", dtype=torch.bfloat16),
              out_f32=torch.empty((M, N), device="cuda"), bias=torch.zeros(N, device="cuda"), resid=torch.zeros((M, N), device="cuda"),
              aux=torch.zeros((M, N), device="cuda", dtype=torch.bfloat16), drop=(1, 2, 0, 0.1))
    if epi == L.EPI_MUL_BF16: kw["colsum_part"] = torch.empty((M // 128, N), device="cuda")
    for _ in range(reps): gemm(A, B, form, epi, M, N, K, **kw)
    torch.cuda.synchronize()
for M, N in [(768, 3072), (3072, 768), (768, 768), (2304, 768)]:
    A, B = rnd(T, M), rnd(T, N)
    sp = lib.carel_gemm_wgrad_splits(M, N, T)
    slabs = torch.empty((sp, M, N), device="cuda")
    for _ in range(reps): gemm(A, B, L.GEMM_TN, L.EPI_SLAB_F32, M, N, T, splits=sp, out_f32=slabs)
    torch.cuda.synchronize()
L.check(lib.carel_gemm_set_variant(0))
