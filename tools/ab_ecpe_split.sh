#!/bin/bash
# ECPE-shaped step: internally split NT / NN GEMMs on the 128x128 kernel (hook 140) vs on the ping-pong kernel (141, default), three rounds
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --no-ecpe --shape B --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-24s %.3f ms/step (median %.3f; GEMM avg %.1f us)' % ('$*' or 'default', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_us']))"; }
for i in 1 2 3; do
  run --gemm-variant 140
  run --gemm-variant 141
done
