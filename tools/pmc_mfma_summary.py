#!/usr/bin/env python3
"""Per-kernel matrix-core occupancy from one rocprofv3 counter_collection.csv:
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is
summed over the 8 XCDs, MI355X_MICROARCH.md); the SQ wait / active buckets are quad-cycle wave counters, reported as fractions of
SQ_WAVE_CYCLES."""
import collections, csv, sys

tot = collections.defaultdict(collections.Counter)
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE")
print("# of: python3 bench.py --no-overlap --no-ecpe --no-cpu-baseline --steps 2 --warmup 1   (dense shape-A steps, one stream)")
print("kernel,launches,kernel_cycles_per_launch,mfma_busy_cycles_per_launch,mfma_util,wait_any_frac,wait_inst_frac,active_inst_frac")
rows = []
for k, c in tot.items():
    n = max(1, cnt[k])
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    wc = max(1.0, c.get("SQ_WAVE_CYCLES", 0.0))
    util = mf / (1024.0 * cyc) if cyc else 0.0
    rows.append((k, n, cyc / n, mf / n, util, c.get("SQ_WAIT_ANY", 0.0) / wc, c.get("SQ_WAIT_INST_ANY", 0.0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, cyc, mf))
rows.sort(key=lambda r: -r[8])
for r in rows:
    print('"%s",%d,%.0f,%.0f,%.4f,%.3f,%.3f,%.3f' % r[:8])
g = [r for r in rows if "gemm_kernel" in r[0] or "gemm_pp_kernel" in r[0]]
if g:
    cyc, mf = sum(r[8] for r in g), sum(r[9] for r in g)
    print('"ALL carel::gemm_pp_kernel + carel::gemm_kernel instantiations",%d,%.0f,%.0f,%.4f,,,' % (sum(r[1] for r in g), cyc / sum(r[1] for r in g), mf / sum(r[1] for r in g), mf / (1024.0 * cyc)))
