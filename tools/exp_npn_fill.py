#!/usr/bin/env python3
"""Is the operand-staging ceiling of the GEMM loops per CU or for the whole chip?  N = 768 GEMMs at M = 8192 with the 256 x 96 tile (256 workgroups, 70 FLOP per
staged byte) against the 256 x 192 tile forced (hook 72: 128 workgroups on 128 CUs, 110 FLOP per staged byte, 0.64x the staged bytes).  Run under
rocprofv3 --kernel-trace (tools/exp_npn_fill.sh): kernel durations come from the trace."""
import os, sys
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
M = 8192
for (N, K) in [(768, 3072), (768, 768), (768, 2304)]:
    A, B = rnd(M, K), rnd(N, K)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    for hook in (70, 72):
        L.check(lib.carel_gemm_set_variant(3)); L.check(lib.carel_gemm_set_variant(hook))
        for _ in range(30): gemm(A, B, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, out_bf16=out, bias=bias)
        torch.cuda.synchronize()
L.check(lib.carel_gemm_set_variant(70)); L.check(lib.carel_gemm_set_variant(0))
