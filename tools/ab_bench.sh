#!/bin/bash
# A/B of bench.py options on one box (same process conditions): prints ms/step per option set, two rounds.
run() { python bench.py --no-cpu-baseline --no-ecpe --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %.3f ms/step  (GEMM avg %.1f us)' % ('$*' or 'default', d['ms_per_step'], d['roofline']['avg_launch_us']))"; }
for i in 1 2; do
  run
  run --no-forward-chains
  run --no-adam-in-backward
  run --no-forward-chains --no-adam-in-backward
  run --no-overlap
done
