cd $GRAFT_REPO_ROOT
: > gpurun_out/off.log
for round in 1 2; do
  for off in 0 256 4096 65536 1048576 1052672; do
    for args in "" "--sorted"; do
      echo "== offset $off $args" >> gpurun_out/off.log
      timeout -k 10 60 python3 tools/sort_bench.py --reps 3 --alt-offset $off $args 2>&1 | grep "^pass:\|rror" >> gpurun_out/off.log
    done
  done
done
cat gpurun_out/off.log
