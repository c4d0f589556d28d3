#!/bin/bash
# Adam launch width against the step: dense and ECPE-shaped ms per step with the Adam grid capped at 32 << k workgroups (hook 280 + k; 292 = one float4
# per thread, the default).  Fused into backward (bench default).
run() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s dense %.3f ms  ecpe %.3f ms' % ('$*', d['ms_per_step'], d['ecpe_shaped']['ms_per_step']))"; }
for i in 1 2; do
  for v in 292 287 285 284 283 282 281 280; do run --gemm-variant $v; done
done
