#!/usr/bin/env python3
"""FFN1 forward (8192 x 3072 x 768, NT) with each epilogue the ping-pong kernel has for it: what the GELU arithmetic and the second
output array cost on top of the bias -> bf16 GEMM.  Measurement only."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
M, N, K = 8192, 3072, 768
g = torch.Generator().manual_seed(0)
SC = float(os.environ.get("WSCALE", 0.05))      # pre-activation std = 0.5 * SC * sqrt(768): 0.7 at the default; WSCALE=0.5 -> 6.9 (2 % of |u| >= 16)
A = (torch.randn((M, K), generator=g) * 0.5).cuda().bfloat16(); B = (torch.randn((N, K), generator=g) * SC).cuda().bfloat16()
o0 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16); o1 = torch.empty_like(o0); bias = torch.zeros(N, device="cuda")
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
cases = [("bias -> bf16 (one output, no GELU)", lambda: gemm(A, B, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, out_bf16=o0, bias=bias)),
         ("bias + GELU, one output (inference)", lambda: gemm(A, B, L.GEMM_NT, L.EPI_BIAS_GELU, M, N, K, out2_bf16=o1, bias=bias)),
         ("bias + GELU, u and g stored", lambda: gemm(A, B, L.GEMM_NT, L.EPI_BIAS_GELU, M, N, K, out_bf16=o0, out2_bf16=o1, bias=bias)),
         ("bias + GELU + gelu', g and g' stored (training)", lambda: gemm(A, B, L.GEMM_NT, L.EPI_BIAS_GELU_DG, M, N, K, out_bf16=o0, out2_bf16=o1, bias=bias))]
for r in range(3):
    for name, fn in cases:
        print("%-52s %6.1f us" % (name, statistics.median(timed(fn) for _ in range(3))), flush=True)
