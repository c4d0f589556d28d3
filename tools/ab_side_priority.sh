#!/bin/bash
# dense / ECPE-shaped step against the priority of the library's side streams (experiments build: CAREL_SIDE_STREAM_PRIORITY = low | normal | high)
run() { CAREL_SIDE_STREAM_PRIORITY=$1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --gemm-variant 292 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s dense %.3f ms  ecpe %.3f ms' % ('$1', d['ms_per_step'], d['ecpe_shaped']['ms_per_step']))"; }
for i in 1 2; do run low; run normal; run high; done
