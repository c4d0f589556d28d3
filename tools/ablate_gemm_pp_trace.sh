cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MS=${MS:-1664} CAREL_HIP_EXP_LIB=carel_vae_amd/libcarel_hip_ablate.so    # build: CAREL_BUILD_TAG=ablate CAREL_EXTRA_FLAGS="-DCAREL_GEMM_ABLATE -DCAREL_EXPERIMENTS" python -m carel_vae_amd.build
rm -rf gpurun_out/prof_abl
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_abl -- python3 tools/ablate_gemm_pp.py > gpurun_out/abl.log 2>&1
cp "$(find gpurun_out/prof_abl -name '*kernel_trace.csv' | head -1)" gpurun_out/abl_trace.csv
rm -rf gpurun_out/prof_abl
python3 - <<'PY'
import csv, collections, re
d = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/abl_trace.csv')):
    n = r['Kernel_Name']
    if 'gemm_pp_kernel' not in n: continue
    g = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) if 'Grid_Size_X' in r else 0
    m = re.search(r'gemm_pp_kernel<([^>]*)>', n)
    d[(m.group(1), g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
import statistics
for k, v in sorted(d.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(k, len(v), "median %.1f us  min %.1f" % (statistics.median(v), min(v)))
PY
