#!/bin/bash
# A/B of two builds of the library on one box: carel_vae_amd/libcarel_hip_old.so vs libcarel_hip_new.so (loaded through CAREL_HIP_LIB in
# turn; the product library is not touched), serial (--no-overlap) and default step, three rounds.  Build the two files with
# `CAREL_BUILD_TAG=old python -m carel_vae_amd.build` / `CAREL_BUILD_TAG=new ...` on the two trees.
cd "$(dirname "$0")/.."
run() { CAREL_HIP_LIB=$PWD/carel_vae_amd/libcarel_hip_$V.so python bench.py --no-cpu-baseline --no-ecpe --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-4s %-14s %.3f ms/step  (GEMM avg %.1f us)' % ('$V', '$*' or 'default', d['ms_per_step'], d['roofline']['avg_launch_us']))"; }
for i in 1 2 3; do
  for V in old new; do
    run
    run --no-overlap
  done
done
