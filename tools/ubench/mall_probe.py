#!/usr/bin/env python3
"""Does a freshly WRITTEN 50-MB buffer read back at Infinity-Cache speed?  (measurement only; torch ops as traffic generators)
Times a streaming read (sum) of y: re-read hot; right after y was written by a copy kernel; after 100 MB of other writes following that
write; after 600 MB of other traffic (cold).  Answers whether the operands a GEMM of the training step reads right after their producer
kernel are warm, as the stand-alone shape bench assumes."""
import torch, statistics
dev = "cuda"
n = 25 * 1024 * 1024                      # 50 MB of bf16
y = torch.empty(n, device=dev, dtype=torch.bfloat16)
z = torch.randn(n, device=dev).bfloat16()
o1, o2 = torch.empty(n, device=dev, dtype=torch.bfloat16), torch.empty(n, device=dev, dtype=torch.bfloat16)
big = torch.empty(300 * 1024 * 1024, device=dev, dtype=torch.bfloat16)     # 600 MB
def t_read(prep, reps=9):
    ts = []
    for _ in range(reps):
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); s = y.float().sum() if False else torch.sum(y, dtype=torch.float32); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
y.copy_(z); torch.cuda.synchronize()
print("read 50 MB hot (re-read):              %.1f us" % t_read(lambda: torch.sum(y, dtype=torch.float32)))
print("read right after it was written:       %.1f us" % t_read(lambda: y.copy_(z)))
def w100(): y.copy_(z); o1.fill_(1.0); o2.fill_(2.0)
print("written, then 100 MB of other writes:  %.1f us" % t_read(w100))
def cold(): y.copy_(z); big.fill_(0.5)
print("written, then 600 MB of other writes:  %.1f us (cold)" % t_read(cold))
def cold_r(): torch.sum(y, dtype=torch.float32); torch.sum(big, dtype=torch.float32)
print("read, then 600 MB of other reads:      %.1f us (cold)" % t_read(cold_r))
