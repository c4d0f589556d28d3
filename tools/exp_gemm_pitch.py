#!/usr/bin/env python3
"""Does the leading dimension of the operands matter?  (measurement only)
The encoder's operand pitches are 1536 / 4608 / 6144 bytes = 12 / 36 / 48 cache lines: the rows of a K tile start 12-48 lines apart, so if the
L2 channel of a line were a plain function of its low address bits, a tile's rows would pile up on 1-4 of an XCD's 16 channels.  Times the
production GEMM shapes with the operands stored at their natural pitch and at a pitch padded by 8 / 64 / 128 / 200 elements."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm

lib = L.load()
T = 8192
g = torch.Generator().manual_seed(0)
def rnd(r, c, pad):
    full = (torch.randn((r, c + pad), generator=g) * 0.5).cuda().bfloat16()
    return full[:, :c] if pad else full
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
shapes = [("fwd FFN2 NT", L.GEMM_NT, L.EPI_BIAS_BF16, T, 768, 3072, 1),
          ("fwd QKV  NT", L.GEMM_NT, L.EPI_BIAS_BF16, T, 2304, 768, 1),
          ("fwd FFN1 NT", L.GEMM_NT, L.EPI_BIAS_BF16, T, 3072, 768, 1),
          ("dgrad FFN1 NN", L.GEMM_NN, L.EPI_BIAS_BF16, T, 768, 3072, 1),
          ("dgrad FFN2 NN", L.GEMM_NN, L.EPI_BIAS_BF16, T, 3072, 768, 1),
          ("wgrad FFN2 TN", L.GEMM_TN, L.EPI_SLAB_F32, 768, 3072, T, 5),
          ("wgrad FFN1 TN", L.GEMM_TN, L.EPI_SLAB_F32, 3072, 768, T, 5),
          ("wgrad QKV  TN", L.GEMM_TN, L.EPI_SLAB_F32, 2304, 768, T, 3)]
pads = [0, 8, 64, 128, 200]
for name, form, epi, M, N, K, sp in shapes:
    res = {}
    for pa in pads:
        for pb in (0, pa) if pa else (0,):
            if form == L.GEMM_NT: A, B = rnd(M, K, pa), rnd(N, K, pb)
            elif form == L.GEMM_NN: A, B = rnd(M, K, pa), rnd(K, N, pb)
            else: A, B = rnd(K, M, pa), rnd(K, N, pb)
            kw = dict(bias=torch.zeros(N, device="cuda"))
            if epi == L.EPI_SLAB_F32: kw = dict(out_f32=torch.empty((sp, M, N), device="cuda"), splits=sp)
            else: kw["out_bf16"] = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
            f = lambda: gemm(A, B, form, epi, M, N, K, **kw)
            f(); ts = [timed(f) for _ in range(5)]
            res[(pa, pb)] = statistics.median(ts)
    print("%-14s M=%5d N=%5d K=%5d | " % (name, M, N, K) + " | ".join("A+%d B+%d %6.1f" % (k[0], k[1], v) for k, v in res.items()), flush=True)
