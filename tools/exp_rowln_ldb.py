#!/usr/bin/env python3
"""Is the fused row-band kernel's weight stream slowed by L2 channel hot-spotting (every workgroup reads the same 16 rows x 64 B at the
same time, row pitch 1536 / 6144 B)?  Same launch with padded weight leading dimensions.  python tools/exp_rowln_ldb.py"""
import os, sys, statistics, ctypes as C
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
lib = L.load()
T = int(os.environ.get("T", 8192))
g0 = torch.Generator().manual_seed(0)
def rnd(*s, sc=0.5): return (torch.randn(s, generator=g0) * sc).cuda()
gamma, beta, bias = 1 + rnd(768, sc=0.1), rnd(768, sc=0.1), rnd(768, sc=0.1)
for K in (768, 3072):
    A, r = rnd(T, K).bfloat16(), rnd(T, 768)
    h, xf, xb, st = torch.empty((T, 768), device="cuda"), torch.empty((T, 768), device="cuda"), torch.empty((T, 768), device="cuda", dtype=torch.bfloat16), torch.empty((T, 2), device="cuda")
    for pad in (0, 8, 64, 72, 128, 200, 520):
        Wp = rnd(768, K + pad, sc=0.03).bfloat16()
        a = L.GemmRowLnArgs()
        a.A, a.W, a.lda, a.ldb, a.M, a.K = A.data_ptr(), Wp.data_ptr(), K, K + pad, T, K
        a.bias, a.resid_f32, a.gamma, a.beta, a.eps = bias.data_ptr(), r.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-12
        a.h_f32, a.x_f32, a.x_bf16, a.stats = h.data_ptr(), xf.data_ptr(), xb.data_ptr(), st.data_ptr()
        a.drop_seed, a.drop_site, a.drop_idx_offset, a.drop_p, a.drop_row_map = 1, 2, 0, 0.1, None
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            L.check(lib.carel_gemm_rowln(C.byref(a), L.current_stream()))
            e0.record()
            for _ in range(20): L.check(lib.carel_gemm_rowln(C.byref(a), L.current_stream()))
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        print("K = %4d  ldb = K + %3d: %6.1f us" % (K, pad, statistics.median(ts)), flush=True)
