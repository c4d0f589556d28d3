#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_npn
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_npn -- python3 tools/exp_npn_fill.py > gpurun_out/npn.log 2>&1
python3 - <<'PY'
import csv, glob, re, statistics, collections
f = glob.glob('gpurun_out/prof_npn/**/*kernel_trace.csv', recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    m = re.search(r'gemm_pp_kernel<([^>]*)>', n)
    if not m: continue
    key = (m.group(1).replace(' ', ''), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
    d.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    # launches come in runs of 30 per (shape, hook): print medians of consecutive runs
    for i in range(0, len(v), 30):
        print(k, "run %d: median %.1f us" % (i // 30, statistics.median(v[i:i + 30])))
PY
rm -rf gpurun_out/prof_npn
