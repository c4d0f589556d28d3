#!/bin/bash
# A/B kernel statistics of the overlapped default step: ping-pong GEMM with the fine schedule (hook 90) vs the wide one (default).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in fine wide; do
  rm -rf gpurun_out/prof_$v
  if [ $v = fine ]; then extra="--gemm-variant 90"; else extra=""; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$v -- python3 bench.py --no-cpu-baseline --no-ecpe --steps 30 --warmup 5 $extra > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  cp "$(find gpurun_out/prof_$v -name '*kernel_stats.csv' | head -1)" gpurun_out/ab_${v}_kernel_stats.csv
  cp "$(find gpurun_out/prof_$v -name '*kernel_trace.csv' | head -1)" gpurun_out/ab_${v}_kernel_trace.csv
  rm -rf gpurun_out/prof_$v
done
