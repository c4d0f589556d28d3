#!/usr/bin/env python3
"""Would the two FFN weight gradients of a layer run faster SIDE BY SIDE at a lower split-K factor than one after the other at the production
factor?  (measurement only)  Each is 48 tiles of 256 x 192 over K = 8192 tokens; today: split 5 (240 workgroups) + slab reduction, twice.
Here: both at split 2 / 3 (96 / 144 workgroups each) on two streams at once, against the sequential production pair."""
import os, sys, statistics, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
lib.carel_side_stream.restype = C.c_void_p
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(s, generator=g) * 0.5).cuda().bfloat16()
T = 8192
dy, gg, du, x1 = rnd(T, 768), rnd(T, 3072), rnd(T, 3072), rnd(T, 768)
dW2, dW1 = torch.empty((768, 3072), device="cuda"), torch.empty((3072, 768), device="cuda")
side = torch.cuda.ExternalStream(lib.carel_side_stream(0))
main = torch.cuda.current_stream()
def wg(A, B, M, N, sp, slabs, dW, stream):
    with torch.cuda.stream(stream):
        gemm(A, B, L.GEMM_TN, L.EPI_SLAB_F32, M, N, T, splits=sp, out_f32=slabs)
        L.check(lib.carel_slab_reduce_f32(slabs.data_ptr(), dW.data_ptr(), M * N, sp, 0, L.current_stream()))
def run(sp2, sp1, concurrent):
    s2, s1 = torch.empty((sp2, 768, 3072), device="cuda"), torch.empty((sp1, 3072, 768), device="cuda")
    def once():
        if concurrent:
            side.wait_stream(main)
            wg(dy, gg, 768, 3072, sp2, s2, dW2, main)
            wg(du, x1, 3072, 768, sp1, s1, dW1, side)
            main.wait_stream(side)
        else:
            wg(dy, gg, 768, 3072, sp2, s2, dW2, main)
            wg(du, x1, 3072, 768, sp1, s1, dW1, main)
    for _ in range(3): once()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): once()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    return statistics.median(ts)
print("sequential, split 5 + 5 (production): %.1f us per pair (GEMM + reduction each)" % run(5, 5, False))
for sp2, sp1 in [(2, 2), (3, 2), (2, 3), (3, 3), (5, 5)]:
    print("side by side, split %d + %d: %.1f us per pair" % (sp2, sp1, run(sp2, sp1, True)), flush=True)
