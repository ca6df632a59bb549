export PYTHONPATH=.
for pt in 2 4 8 16 32; do
  echo "per_thread=$pt $(IGX_QUANTILE_PER_THREAD=$pt timeout -k 10 90 python -u tools/bench_tasks.py ant_imitate 2>/dev/null | grep ^quantile_f64)" || exit 1
done
