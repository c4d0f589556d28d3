#!/usr/bin/env python3
"""Per-launch durations of ONE training step in a rocprofv3 kernel trace of the serial run, in launch order, median over the last
steps of the trace: which GEMM of the layer costs what INSIDE the step (the stand-alone shape bench re-reads operands that sit in
the Infinity Cache).  python tools/trace_step_seq.py trace.csv [gemm|all]"""
import csv, sys, re, statistics, collections
rows = list(csv.DictReader(open(sys.argv[1])))
what = sys.argv[2] if len(sys.argv) > 2 else "gemm"
def nm(r):
    n = r["Kernel_Name"]
    m = re.search(r"gemm_pp_kernel<([^>]*)>", n)
    if m: return "pp<%s>" % m.group(1).replace(" ", "")
    m = re.search(r"gemm_kernel(_big)?<([^>]*)>", n)
    if m: return "v1<%s>" % m.group(2).replace(" ", "")
    return n.split("(")[0].replace("void carel::", "").replace("carel::", "").replace("(anonymous namespace)::", "")[:40]
def grid(r):
    try: return "%dx%dx%d" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    except Exception: return "?"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r), grid(r)) for r in rows)
starts = [i for i, e in enumerate(ev) if e[2].startswith("embed_fwd")]
steps = [ev[a:b] for a, b in zip(starts[-9:-1], starts[-8:])]
n = len(steps[0])
steps = [s for s in steps if len(s) == n]
print("%d steps of %d launches; step %.3f ms busy %.3f ms" % (len(steps), n, statistics.median((s[-1][1] - s[0][0]) for s in steps) / 1e6,
      statistics.median(sum(e[1] - e[0] for e in s) for s in steps) / 1e6))
tot = collections.defaultdict(lambda: [0.0, 0])
for i in range(n):
    name, g = steps[0][i][2], steps[0][i][3]
    d = statistics.median(s[i][1] - s[i][0] for s in steps) / 1e3
    gap = statistics.median((s[i][0] - s[i - 1][1]) for s in steps) / 1e3 if i else 0.0
    tot[(name, g)][0] += d; tot[(name, g)][1] += 1
    if what == "all" or name.startswith(("pp<", "v1<")): print("%4d %-34s %-12s %7.1f us  gap %5.1f" % (i, name, g, d, gap))
print("--- per (kernel, grid)")
for (name, g), (d, c) in sorted(tot.items(), key=lambda x: -x[1][0]):
    print("%-34s %-12s n %3d  avg %7.1f us  sum %8.1f us" % (name, g, c, d / c, d))
