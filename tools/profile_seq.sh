#!/bin/bash
# per-launch sequence of the serial dense step (rocprofv3 kernel trace -> tools/trace_step_seq.py); tools/profile_ecpe_seq.sh is the packed twin
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-rXX}
rm -rf gpurun_out/prof_ds
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ds -- python3 bench.py --no-overlap --no-ecpe --no-cpu-baseline --steps 12 --warmup 4 > gpurun_out/ds.json 2> gpurun_out/ds.err
python3 tools/trace_step_seq.py "$(find gpurun_out/prof_ds -name '*kernel_trace.csv' | head -1)" all > gpurun_out/${tag}_step_seq.txt
rm -rf gpurun_out/prof_ds
head -1 gpurun_out/${tag}_step_seq.txt
