#!/bin/bash
# L2-miss (fabric-side) read bytes per GEMM launch, per shape, for the XCD tile maps of the ping-pong kernel: rocprofv3 --pmc FETCH_SIZE
# (its own pass, kernel trace only) of tools/run_gemm_shapes.py with hook 120 (row-major chunks per XCD) and 121 (rectangles, default).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-rXX}
out=gpurun_out/${tag}_pmc_gemm_fetch.csv
echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/run_gemm_shapes.py 3 4 <hook>; MB per launch = counter KiB x 2 (gfx950 correction, MI355X_MICROARCH.md)" > $out
echo "xcd_map,kernel,grid_x,grid_z,launches,fetch_MB_per_launch_corrected" >> $out
for hook in 120 121; do
  rm -rf gpurun_out/pmc_gf
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_gf -- python3 tools/run_gemm_shapes.py 3 4 $hook > gpurun_out/pmc_gf.log 2>&1
  python3 - "$(find gpurun_out/pmc_gf -name '*counter_collection.csv' | head -1)" $hook >> $out <<'PY'
import csv, sys, collections, re
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm" not in r["Kernel_Name"] or r["Counter_Name"] != "FETCH_SIZE": continue
    m = re.search(r"(gemm\w*kernel<[^>]*>)", r["Kernel_Name"])
    k = (m.group(1) if m else r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Z", ""))
    a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, gx, gz), (n, v) in acc.items():
    print('%s,"%s",%s,%s,%d,%.1f' % ("chunks" if sys.argv[2] == "120" else "rect", k, gx, gz, n, v / n * 1024 * 2 / 1e6))
PY
done
rm -rf gpurun_out/pmc_gf
cat $out
