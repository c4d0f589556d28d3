#!/bin/bash
# Round profile set (run on the GPU box): serial kernel stats that the roofline leg must agree with, the overlapped
# default run, and the default bench line with the CPU baseline.  Outputs land in gpurun_out/ (copy into profiles/).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-rXX}
rm -rf gpurun_out/prof_serial gpurun_out/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial -- python3 bench.py --no-overlap --no-ecpe --no-cpu-baseline > gpurun_out/${tag}_bench_serial_profiled.json 2> gpurun_out/prof_serial.err
cp "$(find gpurun_out/prof_serial -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_bench_serial_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -- python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_default_profiled.json 2> gpurun_out/prof_default.err
cp "$(find gpurun_out/prof_default -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_bench_default_kernel_stats.csv
rm -rf gpurun_out/prof_serial gpurun_out/prof_default
python bench.py > gpurun_out/${tag}_bench_with_cpu_baseline.json 2> gpurun_out/bench_default.err
tail -1 gpurun_out/${tag}_bench_with_cpu_baseline.json | cut -c1-300
