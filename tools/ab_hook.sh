#!/bin/bash
# A/B of one tuning hook on ONE box: bench.py with carel_gemm_set_variant(<off>) applied first against the default, alternating.
# usage: tools/ab_hook.sh <hook number that switches the feature OFF> [rounds] [steps]
hook=$1; rounds=${2:-2}; steps=${3:-40}
for i in $(seq $rounds); do
  # (both legs on the experiments library -- the product library has no hooks: "on" = hook 0 = automatic kernel choice, a no-op)
  python bench.py --gemm-variant $hook --no-cpu-baseline --steps $steps > gpurun_out/b_off.json 2>/dev/null || exit 1
  python bench.py --gemm-variant 0 --no-cpu-baseline --steps $steps > gpurun_out/b_on.json 2>/dev/null || exit 1
  python - <<'PY'
import json
for t in ("off", "on"):
    d = json.loads(open("gpurun_out/b_%s.json" % t).read().strip().splitlines()[-1])
    print("%-4s %.3f ms/step  %.0f pairs/s  ecpe %.3f ms  gemm frac %.4f (kernel only %.4f)" % (t, d["ms_per_step"], d["value"], d.get("ecpe_shaped", {}).get("ms_per_step", 0), d["roofline"]["frac"], d["roofline"].get("frac_kernel_only", 0)))
PY
done
