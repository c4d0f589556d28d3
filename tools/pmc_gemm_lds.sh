#!/bin/bash
# LDS bank conflicts of the GEMM kernels: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (+ kernel trace) of tools/run_gemm_shapes.py;
# conflict share = extra LDS cycles / all LDS-array cycles (MI355X_MICROARCH.md "LDS")
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-rXX}
rm -rf gpurun_out/pmc_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python3 tools/run_gemm_shapes.py 3 4 > gpurun_out/pmc_lds.log 2>&1
python3 - "$(find gpurun_out/pmc_lds -name '*counter_collection.csv' | head -1)" > gpurun_out/${tag}_pmc_gemm_lds.csv <<'PY'
import csv, sys, collections, re
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm" not in r["Kernel_Name"]: continue
    m = re.search(r"(gemm\w*kernel<[^>]*>)", r["Kernel_Name"])
    k = (m.group(1) if m else r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")))
    a = acc.setdefault(k, collections.defaultdict(float)); a[r["Counter_Name"]] += float(r["Counter_Value"]); a["n"] += 1
print("kernel,grid_x,launches,lds_bank_conflict_cycles,lds_idx_active_cycles,conflict_share,lds_insts")
for (k, gx), a in acc.items():
    n = a["n"] / 4
    print('"%s",%s,%d,%.0f,%.0f,%.4f,%.0f' % (k, gx, n, a["SQ_LDS_BANK_CONFLICT"] / n, a["SQ_LDS_IDX_ACTIVE"] / n, a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1), a["SQ_INSTS_LDS"] / n))
PY
rm -rf gpurun_out/pmc_lds
cat gpurun_out/${tag}_pmc_gemm_lds.csv
