#!/usr/bin/env python3
"""carel_sgemm_f32 (f32-input MFMA) on the shapes of the English model's vocabulary-wide heads: time per launch, TFLOP/s against
the 157.3 TFLOP/s f32 matrix peak and GB/s of algorithmic bytes against 8 TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
lib = L.load()
st = L.current_stream()
B, V = 64, 22463


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for K in (24, 384, 432):
    x = torch.randn(B, K, device="cuda"); W = torch.randn(V, K, device="cuda"); bias = torch.randn(V, device="cuda")
    Lg = torch.empty(B, V, device="cuda"); dW = torch.empty(V, K, device="cuda")
    sp = 64
    parts = torch.empty(sp, B, K, device="cuda")
    t1 = timeit(lambda: L.check(lib.carel_sgemm_f32(x.data_ptr(), K, 0, W.data_ptr(), K, 0, Lg.data_ptr(), V, B, V, K, bias.data_ptr(), 0, 1, 0, st)))
    t2 = timeit(lambda: L.check(lib.carel_sgemm_f32(Lg.data_ptr(), V, 1, x.data_ptr(), K, 1, dW.data_ptr(), K, V, K, B, None, 0, 1, 0, st)))
    t3 = timeit(lambda: L.check(lib.carel_sgemm_f32(Lg.data_ptr(), V, 0, W.data_ptr(), K, 1, parts.data_ptr(), K, B, K, V, None, 0, sp, B * K, st)))
    fl = 2.0 * B * V * K
    by1 = 4.0 * (V * K + B * K + B * V); by2 = 4.0 * (B * V + B * K + V * K); by3 = 4.0 * (B * V + V * K + sp * B * K)
    for name, t, by in (("logits  x W^T ", t1, by1), ("dW = dL^T x   ", t2, by2), ("dx = dL W /64 ", t3, by3)):
        print("K=%3d %s %6.1f us  %5.1f TF (%.0f %% of 157.3)  %5.2f TB/s alg. (%.0f %% of 8)" % (K, name, t, fl / t / 1e6, 100 * fl / t / 1e6 / 157.3, by / t / 1e6, 100 * by / t / 1e6 / 8))
