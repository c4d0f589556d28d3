#!/usr/bin/env python3
"""Attention forward / backward kernels stand-alone (B = 64, S = 128, 12 heads x 64): time per launch with and without dropout
on the probabilities, and the MFMA rate they reach (forward 4 B NH S^2 d flop, backward 10 B NH S^2 d)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
lib = L.load()
B, S, NH, HD = 64, 128, 12, 64
H = NH * HD
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * S, 3 * H, generator=g) * 0.5).cuda().bfloat16()
dctx = (torch.randn(B * S, H, generator=g) * 0.1).cuda().bfloat16()
ctx = torch.empty((B * S, H), device="cuda", dtype=torch.bfloat16)
lse = torch.empty((B, NH, S), device="cuda")
dqkv = torch.empty((B * S, 3 * H), device="cuda", dtype=torch.bfloat16)


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for p in (0.0, 0.1):
    a = L.AttnArgs()
    a.qkv, a.attention_mask, a.ctx, a.lse, a.dctx, a.dqkv = qkv.data_ptr(), None, ctx.data_ptr(), lse.data_ptr(), dctx.data_ptr(), dqkv.data_ptr()
    a.batch, a.seq_len, a.heads, a.head_dim = B, S, NH, HD
    a.drop_seed, a.drop_site, a.drop_idx_offset, a.drop_p = 5, 1, 0, p
    st = L.current_stream()
    tf = timeit(lambda: L.check(lib.carel_attention_fwd(C.byref(a), st)))
    tb = timeit(lambda: L.check(lib.carel_attention_bwd(C.byref(a), st)))
    ff, fb = 4.0 * B * NH * S * S * HD, 10.0 * B * NH * S * S * HD
    print("dropout %.1f: fwd %5.1f us (%4.0f TF)   bwd %5.1f us (%4.0f TF)" % (p, tf, ff / tf / 1e6, tb, fb / tb / 1e6))
