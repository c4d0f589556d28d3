#!/bin/bash
# kernel trace of the serial step (one stream): for gap analysis (tools/trace_gaps.py)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_ser
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ser -- python3 bench.py --no-cpu-baseline --no-ecpe --no-overlap --steps 20 --warmup 5 > gpurun_out/ser.json 2> gpurun_out/ser.err
cp "$(find gpurun_out/prof_ser -name '*kernel_trace.csv' | head -1)" gpurun_out/serial_kernel_trace.csv
rm -rf gpurun_out/prof_ser
