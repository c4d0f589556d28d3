#!/bin/bash
# One GPU round: run the -m gpu tests (single process), log to gpurun_out/.
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --timeout=300 "$@" > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -40 gpurun_out/pytest_gpu.log
exit $rc
