#!/bin/bash
# the round's final gate on a GPU box: every -m gpu test, smoke(), the driver's bench command
set -e
python -m pytest tests -m gpu -x -q > gpurun_out/gate_tests.log 2>&1 || { tail -20 gpurun_out/gate_tests.log; exit 1; }
tail -2 gpurun_out/gate_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/gate_smoke.log 2>&1 || { tail -20 gpurun_out/gate_smoke.log; exit 1; }
tail -1 gpurun_out/gate_smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/gate_bench.json 2> gpurun_out/gate_bench.err
tail -1 gpurun_out/gate_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['ecpe_shaped']['ms_per_step'], d['cpu_baseline']['value'])"
