#!/bin/bash
# A/B of bench.py options on one box (same process conditions): prints ms/step per option set, three rounds.
run() { python bench.py --no-cpu-baseline --no-ecpe --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-46s %.3f ms/step  (GEMM avg %.1f us)' % ('$*' or 'default', d['ms_per_step'], d['roofline']['avg_launch_us']))"; }
for i in 1 2 3; do
  run
  run --forward-chains
  run --adam-in-backward
  run --forward-chains --adam-in-backward
  run --no-overlap
done
