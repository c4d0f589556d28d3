#!/bin/bash
# Matrix-core occupancy of every kernel of a serial bench step: one rocprofv3 --pmc pass (SQ counters + GRBM_GUI_ACTIVE),
# summarised by tools/pmc_mfma_summary.py.  Run on the GPU box; result in gpurun_out/<tag>_pmc_mfma_util.csv.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-rXX}
rm -rf gpurun_out/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --no-overlap --no-ecpe --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/pmc_mfma.log 2>&1
python tools/pmc_mfma_summary.py "$(find gpurun_out/pmc_mfma -name '*counter_collection.csv' | head -1)" > gpurun_out/${tag}_pmc_mfma_util.csv
rm -rf gpurun_out/pmc_mfma
head -24 gpurun_out/${tag}_pmc_mfma_util.csv
