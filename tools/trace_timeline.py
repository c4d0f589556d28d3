#!/usr/bin/env python3
"""Per-queue busy time and union-busy fraction of a rocprofv3 kernel trace (steady-state window)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows), key=lambda x: x[0])
ks = ks[int(len(ks) * 0.5):]
t0, t1 = ks[0][0], max(k[1] for k in ks)
span = t1 - t0
# union busy
ev = []
for s, e, *_ in ks:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; depth = 0; last = t0; conc = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last
    conc[depth] += t - last
    depth += d; last = t
print("span %.2f ms, union busy %.2f ms (%.1f%%), idle %.2f ms" % (span / 1e6, busy / 1e6, 100 * busy / span, (span - busy) / 1e6))
for d in sorted(conc): print("  %d kernels in flight: %6.2f ms (%.1f%%)" % (d, conc[d] / 1e6, 100 * conc[d] / span))
perq = collections.defaultdict(int); perqn = collections.Counter()
for s, e, n, q, st in ks:
    perq[(q, st)] += e - s; perqn[(q, st)] += 1
for k, v in sorted(perq.items(), key=lambda x: -x[1]): print("queue/stream %s: busy %.2f ms (%.1f%%), %d kernels" % (k, v / 1e6, 100 * v / span, perqn[k]))
