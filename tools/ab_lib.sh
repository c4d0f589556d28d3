#!/bin/bash
# A/B of two builds of the library on one box: carel_vae_amd/libcarel_hip_old.so vs libcarel_hip_new.so (copied over libcarel_hip.so
# in turn), serial (--no-overlap) and default step, three rounds.  Build the two files with `python -m carel_vae_amd.build` on the two trees.
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --no-ecpe --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-4s %-14s %.3f ms/step  (GEMM avg %.1f us)' % ('$V', '$*' or 'default', d['ms_per_step'], d['roofline']['avg_launch_us']))"; }
for i in 1 2 3; do
  for V in old new; do
    cp carel_vae_amd/libcarel_hip_$V.so carel_vae_amd/libcarel_hip.so
    run
    run --no-overlap
  done
done
cp carel_vae_amd/libcarel_hip_new.so carel_vae_amd/libcarel_hip.so
