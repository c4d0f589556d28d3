#!/bin/bash
# ECPE-shaped step: attention-output forward on the ping-pong kernel directly (330) / split along K in 2 (332) or 3 (333) slabs with the epilogue inside the LayerNorm kernel
run() { python bench.py --no-cpu-baseline --no-ecpe --shape B --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %.3f ms/step  (GEMM avg %.1f us, %.0f TF)' % ('$*' or 'default', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['achieved']))"; }
for i in 1 2; do
  run --gemm-variant 330
  run --gemm-variant 332
  run --gemm-variant 333
done
