#!/usr/bin/env python3
"""Idle gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
# steady-state window: the last 60 % of the trace
lo = int(len(ks) * 0.4)
ks = ks[lo:]
busy = sum(e - s for s, e, _ in ks)
span = ks[-1][1] - ks[0][0]
gaps = collections.Counter(); gsum = collections.Counter()
tot_gap = 0
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    g = s1 - e0
    if g > 0:
        tot_gap += g
        key = n0.split("(")[0][:40] + " -> " + n1.split("(")[0][:40]
        gaps[key] += 1; gsum[key] += g
print("kernels %d  span %.3f ms  busy %.3f ms (%.1f%%)  idle between kernels %.3f ms (%.1f%%)" % (len(ks), span / 1e6, busy / 1e6, 100 * busy / span, tot_gap / 1e6, 100 * tot_gap / span))
print("mean gap %.2f us" % (tot_gap / max(1, len(ks) - 1) / 1e3))
for k, v in gsum.most_common(25):
    print("%8.1f us total  %5d x  %6.2f us  %s" % (v / 1e3, gaps[k], v / gaps[k] / 1e3, k))
