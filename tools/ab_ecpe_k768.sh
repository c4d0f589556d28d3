#!/bin/bash
# ECPE-shaped step against the ping-pong kernel's minimum tile count for K <= 768 GEMMs (hook 40 + k: 32 k tiles)
run() { python bench.py --no-cpu-baseline --no-ecpe --shape B --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %.3f ms/step  (GEMM avg %.1f us, %.0f TF)' % ('$*' or 'default', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['achieved']))"; }
for i in 1 2; do
  run --gemm-variant 43
  run --gemm-variant 42
  run --gemm-variant 41
done
