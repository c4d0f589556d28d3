#!/bin/bash
# Per-shape counters of the GEMM kernels alone (tools/run_gemm_shapes.py): one --pmc pass with SQ counters + GRBM_GUI_ACTIVE and
# the kernel trace, so that cycles / duration gives the clock the chip held.  Run on the GPU box.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-rXX}; variant=${2:-3}
rm -rf gpurun_out/pmc_gemm
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_gemm -- python3 tools/run_gemm_shapes.py $variant 6 > gpurun_out/pmc_gemm.log 2>&1
python3 tools/pmc_gemm_summary.py "$(find gpurun_out/pmc_gemm -name '*counter_collection.csv' | head -1)" "$(find gpurun_out/pmc_gemm -name '*kernel_trace.csv' | head -1)" > gpurun_out/${tag}_pmc_gemm_v${variant}.csv
rm -rf gpurun_out/pmc_gemm
cat gpurun_out/${tag}_pmc_gemm_v${variant}.csv
