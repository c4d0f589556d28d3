#!/bin/bash
# forward slab epilogue fused into the LayerNorm (round 4): model tests, then ECPE-shaped step with the fusion on (271) / off (270) / on + K = 768 split (131)
python -m pytest tests/test_gpu_model.py -m gpu -x -q > gpurun_out/t_model.log 2>&1; tail -3 gpurun_out/t_model.log
run() { python bench.py --no-cpu-baseline --no-ecpe --shape B --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %.3f ms/step' % ('$*' or 'default', d['ms_per_step']))"; }
for i in 1 2; do run --gemm-variant 271; run --gemm-variant 270; run --gemm-variant 271 --gemm-variant 131; done
