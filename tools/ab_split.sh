#!/bin/bash
# internal split-K of the small-grid forward / dgrad GEMMs with K = 768 (hook 130: off, 131: on = default): dense and ECPE-shaped step
run() { python bench.py --no-cpu-baseline --no-ecpe --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-34s %.3f ms/step  (GEMM avg %.1f us)' % ('$*' or 'default', d['ms_per_step'], d['roofline']['avg_launch_us']))"; }
for i in 1 2 3; do
  run --shape A --gemm-variant 130
  run --shape A
  run --shape B --gemm-variant 130
  run --shape B
done
