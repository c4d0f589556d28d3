#!/bin/bash
# A/B of two builds of the library on ONE box: bench.py with CAREL_HIP_LIB = libcarel_hip_<tag>.so (built with CAREL_BUILD_TAG=<tag> from another
# revision) against the product library, alternating, N rounds.  usage: tools/ab_bench_libs.sh <tag> [rounds] [steps]
tag=${1:-base}; rounds=${2:-2}; steps=${3:-40}
for i in $(seq $rounds); do
  CAREL_HIP_LIB=$PWD/carel_vae_amd/libcarel_hip_$tag.so python bench.py --no-cpu-baseline --steps $steps > gpurun_out/b_$tag.json 2>/dev/null || exit 1
  python bench.py --no-cpu-baseline --steps $steps > gpurun_out/b_new.json 2>/dev/null || exit 1
  python - $tag <<'PY'
import json, sys
for t in (sys.argv[1], "new"):
    d = json.loads(open("gpurun_out/b_%s.json" % t).read().strip().splitlines()[-1])
    print("%-6s %.3f ms/step  %.0f pairs/s  ecpe %.3f ms  gemm frac %.4f (kernel only %.4f)" % (t, d["ms_per_step"], d["value"], d.get("ecpe_shaped", {}).get("ms_per_step", 0), d["roofline"]["frac"], d["roofline"].get("frac_kernel_only", 0)))
PY
done
