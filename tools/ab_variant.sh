#!/bin/bash
# A/B of one tuning hook of the GEMM library (carel_gemm_set_variant(V)) against the default, default and serial step, three rounds.
# usage: tools/ab_variant.sh V [extra bench.py flags]
cd "$(dirname "$0")/.."
V=$1; shift
run() { python bench.py --no-cpu-baseline --no-ecpe --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-36s %.3f ms/step  (GEMM avg %.1f us)' % ('$*' or 'default', d['ms_per_step'], d['roofline']['avg_launch_us']))"; }
for i in 1 2 3; do
  run "$@"
  run --gemm-variant $V "$@"
  run --no-overlap "$@"
  run --no-overlap --gemm-variant $V "$@"
done
