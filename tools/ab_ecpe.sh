#!/bin/bash
# ECPE-shaped step (bench.py --shape B) against the ping-pong kernel's minimum tile count (hook 50 + k: 32 k tiles; default 6 = 192)
run() { python bench.py --no-cpu-baseline --no-ecpe --shape B --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %.3f ms/step  (GEMM avg %.1f us, %.0f TF)' % ('$*' or 'default', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['achieved']))"; }
for i in 1 2; do
  run
  run --gemm-variant 55
  run --gemm-variant 54
  run --gemm-variant 53
  run --gemm-variant 52
  run --gemm-variant 51
done
