#!/bin/bash
# ECPE-shaped step under rocprofv3 --kernel-trace, overlapped (default) and serial: which kernels stretch when the side stream is at work?
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SHAPE=${SHAPE:-B}
for mode in overlap serial; do
  rm -rf gpurun_out/prof_eo
  extra=""; [ $mode = serial ] && extra="--no-overlap"
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_eo -- python3 bench.py --shape $SHAPE $extra --no-ecpe --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/eo_$mode.json 2> gpurun_out/eo_$mode.err
  cp "$(find gpurun_out/prof_eo -name '*kernel_trace.csv' | head -1)" gpurun_out/eo_${mode}_trace.csv
done
rm -rf gpurun_out/prof_eo
python3 tools/trace_timeline.py gpurun_out/eo_overlap_trace.csv | tail -12
python3 - <<'PY'
import csv, collections, re
def load(f):
    d = collections.defaultdict(list)
    rows = list(csv.DictReader(open(f)))
    rows = rows[len(rows) // 3:]
    for r in rows:
        n = r['Kernel_Name'].replace('void ', '').replace('carel::', '').replace('(anonymous namespace)::', ''); n = re.sub(r'\((unsigned|carel|float|HIP|long|int).*', '', n)
        d[(n[:70], r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return d
a, b = load('gpurun_out/eo_overlap_trace.csv'), load('gpurun_out/eo_serial_trace.csv')
tot_a = tot_b = 0
out = []
for k in a:
    if k in b:
        sa, sb = sum(a[k]) / len(a[k]), sum(b[k]) / len(b[k])
        out.append((sum(a[k]) - sum(b[k]) * len(a[k]) / len(b[k]), k, len(a[k]), sa, sb))
for d, k, n, sa, sb in sorted(out, reverse=True)[:25]:
    print("%-72s grid %-8s n %5d  overlap %7.1f us  serial %7.1f us  (+%.0f us total)" % (k[0], k[1], n, sa, sb, d))
PY
