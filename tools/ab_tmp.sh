#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1; tail -3 gpurun_out/t_all.log
for i in 1 2; do python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dense %.3f ms  ecpe %.3f ms' % (d['ms_per_step'], d['ecpe_shaped']['ms_per_step']))"; done
bash tools/profile_ecpe_seq.sh r04c > /dev/null 2>&1; grep -n "rowvec\|sum_parts" gpurun_out/r04c_ecpe_step_seq.txt | head -12
