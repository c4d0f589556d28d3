#!/usr/bin/env python3
"""Weight-gradient GEMM + slab reduction at K = T tokens: the 128x128 kernel with its own split factor against the ping-pong
kernel at every split-K factor (hook 100 + s) -- the data behind gemm_pp_wgrad_splits()."""
import os, sys, statistics
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
T = int(os.environ.get("T", 8192))
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, M, N in [("out 768x768", 768, 768), ("QKV 2304x768", 2304, 768), ("FFN2 768x3072", 768, 3072), ("FFN1 3072x768", 3072, 768)]:
    A, B = rnd(T, M), rnd(T, N)
    dW = torch.empty((M, N), device="cuda")
    slabs = torch.empty((16, M, N), device="cuda")
    def run(sp):
        def f():
            gemm(A, B, L.GEMM_TN, L.EPI_SLAB_F32, M, N, T, splits=sp, out_f32=slabs[:sp])
            L.check(lib.carel_slab_reduce_f32(slabs.data_ptr(), dW.data_ptr(), M * N, sp, 0, L.current_stream()))
        f(); return statistics.median(timed(f) for _ in range(4))
    L.check(lib.carel_gemm_set_variant(1)); L.check(lib.carel_gemm_set_variant(100))
    sp1 = lib.carel_gemm_wgrad_splits(M, N, T)
    out = ["128x128/s%d %.1f" % (sp1, run(sp1))]
    L.check(lib.carel_gemm_set_variant(3))
    for sp in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
        L.check(lib.carel_gemm_set_variant(100 + sp))
        if lib.carel_gemm_wgrad_splits(M, N, T) != sp: continue
        out.append("pp/s%d %.1f" % (sp, run(sp)))
    L.check(lib.carel_gemm_set_variant(100)); L.check(lib.carel_gemm_set_variant(0))
    print("%-14s T=%d | %s" % (name, T, " | ".join(out)), flush=True)
