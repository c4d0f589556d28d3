#!/usr/bin/env python3
"""Attention forward / backward kernel times at the bench shape (B = 64, S = 128, 12 heads, dropout 0.1), operands cold (a 256-MB
sweep between launches, as in the training step).  With a library built with CAREL_EXTRA_FLAGS=-DCAREL_ATTN_ABLATE=n the backward
number is that ablation's (1 no dropout hash, 2 no dQ phase, 3 no exp, 4 no stores)."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
lib = L.load()
B, S, NH, H = 64, 128, 12, 768
g = torch.Generator().manual_seed(0)
qkv = (torch.randn((B * S, 3 * H), generator=g)).cuda().bfloat16()
dctx = torch.randn((B * S, H), generator=g).cuda().bfloat16()
ctx = torch.empty((B * S, H), device="cuda", dtype=torch.bfloat16); lse = torch.empty((B, NH, S), device="cuda")
dqkv = torch.empty((B * S, 3 * H), device="cuda", dtype=torch.bfloat16)
a = L.AttnArgs()
a.qkv, a.ctx, a.lse, a.dctx, a.dqkv = qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), dctx.data_ptr(), dqkv.data_ptr()
a.batch, a.seq_len, a.heads, a.head_dim = B, S, NH, 64
a.drop_seed, a.drop_site, a.drop_idx_offset, a.drop_p = 1, 2, 0, 0.1
junk = torch.empty(256 << 20, device="cuda", dtype=torch.uint8)
def timed(fn, n=20, cold=True):
    ts = []
    for _ in range(n):
        if cold: junk.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
fwd = lambda: L.check(lib.carel_attention_fwd(C.byref(a), L.current_stream()))
bwd = lambda: L.check(lib.carel_attention_bwd(C.byref(a), L.current_stream()))
fwd(); bwd()
print("%s: fwd %.1f us cold / %.1f warm   bwd %.1f us cold / %.1f warm" % (os.environ.get("TAG", ""), timed(fwd), timed(fwd, cold=False), timed(bwd), timed(bwd, cold=False)), flush=True)
