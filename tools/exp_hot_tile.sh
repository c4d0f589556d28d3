#!/bin/bash
# Do the ping-pong kernel's K tiles wait for FIRST-TOUCH misses (an operand slice one of the workgroups sharing it must pull from the Infinity Cache)?  The same GEMMs with
# every K tile re-reading the slice's first tile (build: CAREL_BUILD_TAG=hot CAREL_EXTRA_FLAGS="-DCAREL_EXPERIMENTS -DCAREL_PP_HOT_TILE" python -m carel_vae_amd.build)
# against the normal experiments library; kernel-trace durations.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in exp hot; do
  rm -rf gpurun_out/prof_hot
  CAREL_HIP_EXP_LIB=carel_vae_amd/libcarel_hip_$lib.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_hot -- python3 tools/exp_npn_fill.py > gpurun_out/hot.log 2>&1
  echo "== libcarel_hip_$lib.so"
  python3 - <<'PY'
import csv, glob, re, statistics, collections
f = glob.glob('gpurun_out/prof_hot/**/*kernel_trace.csv', recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    m = re.search(r'gemm_pp_kernel<([^>]*)>', r['Kernel_Name'])
    if not m: continue
    key = (m.group(1).replace(' ', ''), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
    d.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    print(k, " | ".join("%.1f" % statistics.median(v[i:i + 30]) for i in range(0, len(v), 30)), "us  (K = 3072, 768, 2304)")
PY
done
rm -rf gpurun_out/prof_hot
