set -e
bash tools/profile_round.sh r02_d > gpurun_out/profile_round.log 2>&1
echo "profile_round done"
bash tools/pmc_traffic.sh r02 > gpurun_out/pmc_traffic.log 2>&1
echo "pmc_traffic done"
bash tools/pmc_mfma.sh r02 > gpurun_out/pmc_mfma.log 2>&1
echo "pmc_mfma done"
bash tools/pmc_gemm_pp.sh r02_d 3 > gpurun_out/pmc_gemm_pp.log 2>&1
echo "pmc_gemm done"
python tools/bench_gemm_pp.py > gpurun_out/r02_d_bench_gemm_pp.txt 2>&1
echo "bench_gemm done"
