#!/bin/bash
# per-launch sequence of the serial ECPE-shaped step (rocprofv3 kernel trace -> tools/trace_step_seq.py)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-rXX}
rm -rf gpurun_out/prof_es
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_es -- python3 bench.py --shape B --no-overlap --no-ecpe --no-cpu-baseline --steps 12 --warmup 4 > gpurun_out/es.json 2> gpurun_out/es.err
python3 tools/trace_step_seq.py "$(find gpurun_out/prof_es -name '*kernel_trace.csv' | head -1)" all > gpurun_out/${tag}_ecpe_step_seq.txt
rm -rf gpurun_out/prof_es
head -3 gpurun_out/${tag}_ecpe_step_seq.txt
