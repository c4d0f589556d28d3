#!/bin/bash
# ECPE-shaped step (bench.py --shape B: ~1.8 k attended tokens per batch of 64 pairs) against the stream options, two rounds.
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --no-ecpe --shape B --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s %.3f ms/step (median %.3f)' % ('$*' or 'default', d['ms_per_step'], d['ms_per_step_median']))"; }
for i in 1 2; do
  run
  run --adam-in-backward
  run --forward-chains
  run --forward-chains --adam-in-backward
  run --no-overlap
done
