#!/bin/bash
# (round 4) A/B on ONE box, alternating: the fused Adam in optim.step() (--adam-in-step) | per layer inside backward() on the auxiliary stream |
# per layer inside backward() queued on the weight-gradient side stream.  Dense and ECPE-shaped step (resident inputs + headline).
rounds=${1:-2}
for i in $(seq $rounds); do
  for v in "default:--adam-in-step" "aux:" "side:"; do
    name=${v%%:*}; flag=${v#*:}
    if [ $name = aux ]; then export CAREL_ADAM_STREAM=1; else unset CAREL_ADAM_STREAM; fi
    python bench.py --no-cpu-baseline --steps 40 $flag > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || exit 1
    python - $name <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("%-8s headline %.3f ms  resident %.3f ms  ecpe %.3f ms" % (sys.argv[1], d["ms_per_step"], d["resident_inputs"]["ms_per_step"], d["ecpe_shaped"]["ms_per_step"]), flush=True)
PY
  done
done
