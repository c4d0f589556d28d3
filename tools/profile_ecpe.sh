#!/bin/bash
# serial kernel statistics of the ECPE-shaped step (packed batches, ~1.8 k tokens): rocprofv3 --kernel-trace --stats of bench.py --shape B
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-rXX}
rm -rf gpurun_out/prof_ecpe
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ecpe -- python3 bench.py --shape B --no-overlap --no-ecpe --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/${tag}_bench_ecpe_serial_profiled.json 2> gpurun_out/prof_ecpe.err
cp "$(find gpurun_out/prof_ecpe -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_bench_ecpe_serial_kernel_stats.csv
rm -rf gpurun_out/prof_ecpe
