#!/bin/bash
# A/B of one tuning hook on ONE box: bench.py with carel_gemm_set_variant(<off>) applied first against the default, alternating.
# usage: tools/ab_hook.sh <hook number that switches the feature OFF> [rounds] [steps]
hook=$1; rounds=${2:-2}; steps=${3:-40}
for i in $(seq $rounds); do
  python - $hook --no-cpu-baseline --steps $steps > gpurun_out/b_off.json 2>/dev/null <<'PY' || exit 1
import sys, runpy
from carel_vae_amd import _lib as L
L.check(L.load().carel_gemm_set_variant(int(sys.argv[1])))
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
PY
  python bench.py --no-cpu-baseline --steps $steps > gpurun_out/b_on.json 2>/dev/null || exit 1
  python - <<'PY'
import json
for t in ("off", "on"):
    d = json.loads(open("gpurun_out/b_%s.json" % t).read().strip().splitlines()[-1])
    print("%-4s %.3f ms/step  %.0f pairs/s  ecpe %.3f ms  gemm frac %.4f (kernel only %.4f)" % (t, d["ms_per_step"], d["value"], d.get("ecpe_shaped", {}).get("ms_per_step", 0), d["roofline"]["frac"], d["roofline"].get("frac_kernel_only", 0)))
PY
done
