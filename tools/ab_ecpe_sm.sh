#!/bin/bash
# ECPE-shaped step with the small-M kernel (gemm_sm.hip) off (340) / on (342); then the serial per-launch sequence with it on
run() { python bench.py --no-cpu-baseline --no-ecpe --shape B --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %.3f ms/step  (GEMM avg %.1f us, %.0f TF)' % ('$*' or 'default', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['achieved']))"; }
for i in 1 2; do run --gemm-variant 340; run --gemm-variant 342; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_es
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_es -- python3 bench.py --shape B --no-overlap --no-ecpe --no-cpu-baseline --steps 12 --warmup 4 --gemm-variant 342 > gpurun_out/es.json 2> gpurun_out/es.err
python3 tools/trace_step_seq.py "$(find gpurun_out/prof_es -name '*kernel_trace.csv' | head -1)" all > gpurun_out/sm_ecpe_step_seq.txt
rm -rf gpurun_out/prof_es
head -1 gpurun_out/sm_ecpe_step_seq.txt; sed -n 150,166p gpurun_out/sm_ecpe_step_seq.txt; sed -n 4,12p gpurun_out/sm_ecpe_step_seq.txt
