run() { python bench.py --no-cpu-baseline --no-ecpe --shape B --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %.3f ms/step  (GEMM avg %.1f us, %.0f TF)' % ('$*' or 'default', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['achieved']))"; }
for i in 1 2; do
  run
  run --gemm-variant 131
  run --gemm-variant 56
done
