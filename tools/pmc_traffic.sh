#!/bin/bash
# HBM-side traffic per kernel launch: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short serial bench run,
# summarised by tools/pmc_summary.py.  Run on the GPU box; result in gpurun_out/<tag>_pmc_hbm_traffic.csv.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-rXX}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --no-overlap --no-ecpe --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/pmc_$c.log 2>&1
done
python tools/pmc_summary.py "$(find gpurun_out/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find gpurun_out/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" > gpurun_out/${tag}_pmc_hbm_traffic.csv
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
head -20 gpurun_out/${tag}_pmc_hbm_traffic.csv
