#!/usr/bin/env python3
"""Per (kernel, grid) summary of one rocprofv3 --pmc + --kernel-trace pass: duration, kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs),
the clock that implies, matrix-core busy share of the 1024 SIMDs, and the SQ wave-time buckets."""
import collections, csv, sys
cc, kt = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Z", ""))
disp = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    disp[r["Dispatch_Id"]][r["Counter_Name"]] = disp[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.defaultdict(list)
for d, c in disp.items():
    if d not in dur: continue
    ns, name, gx, gz = dur[d]
    agg[(name, gx, gz)].append((ns, c))
print("kernel,grid_x,grid_z,launches,avg_us,kernel_cycles,clock_ghz,mfma_util,wait_any,wait_inst,active_inst,valu_insts_per_wave_M")
for (name, gx, gz), v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    v = v[1:] if len(v) > 2 else v           # drop the first (cold) launch
    n = len(v)
    ns = sum(x[0] for x in v) / n
    g = lambda k: sum(x[1].get(k, 0.0) for x in v) / n
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    wc = max(1.0, g("SQ_WAVE_CYCLES"))
    print('"%s",%s,%s,%d,%.1f,%.0f,%.2f,%.3f,%.3f,%.3f,%.3f,%.2f' % (name[:70], gx, gz, n, ns / 1e3, cyc, cyc / ns if ns else 0, g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc) if cyc else 0,
          g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_INSTS_VALU") / 1e6))
