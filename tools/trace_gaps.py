#!/usr/bin/env python3
"""Idle time between consecutive kernels of one training step in a rocprofv3 kernel trace (serial run: one stream, so every gap
is GPU idle time): total, and grouped by the kernel that FOLLOWS the gap.  python tools/trace_gaps.py trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def nm(r): return r["Kernel_Name"].split("(")[0].replace("void carel::", "").replace("carel::", "").replace("(anonymous namespace)::", "")[:44]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r)) for r in rows)
starts = [e[0] for e in ev if e[2].startswith("embed_fwd")]
s0, s1 = starts[-4], starts[-3]
step = [e for e in ev if s0 <= e[0] < s1]
busy = sum(e[1] - e[0] for e in step)
print("step %.3f ms: %d kernels, busy %.3f ms, idle %.3f ms" % ((s1 - s0) / 1e6, len(step), busy / 1e6, (s1 - s0 - busy) / 1e6))
by = collections.defaultdict(lambda: [0, 0])
bk = collections.defaultdict(lambda: [0, 0])
for a, b in zip(step, step[1:]):
    g = max(0, b[0] - a[1]); by[b[2]][0] += g; by[b[2]][1] += 1
for e in step: bk[e[2]][0] += e[1] - e[0]; bk[e[2]][1] += 1
print("%-46s %6s %9s %9s | %9s %9s" % ("kernel", "n", "busy us", "avg us", "gap-before us", "avg gap"))
for k, (d, c) in sorted(bk.items(), key=lambda x: -x[1][0] - by[x[0]][0]):
    print("%-46s %6d %9.1f %9.1f | %9.1f %9.1f" % (k, c, d / 1e3, d / c / 1e3, by[k][0] / 1e3, by[k][0] / max(1, by[k][1]) / 1e3))
