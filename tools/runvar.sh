cd $GRAFT_REPO_ROOT
echo "== full parity" > gpurun_out/variants.log
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 >> gpurun_out/variants.log
for args in "--n 3e5" "--n 6e4"; do
  for lm in 1 100000000000; do
    echo "== $args large_min=$lm" >> gpurun_out/variants.log
    CSTONE_SORT_LARGE_MIN=$lm timeout -k 10 100 python3 tools/sort_bench.py --reps 2 $args 2>&1 | tail -2 >> gpurun_out/variants.log
  done
done
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline >> gpurun_out/variants.log 2>&1
