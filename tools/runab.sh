cd $GRAFT_REPO_ROOT
# A/B on one box: alternate default build and named variants, 2 rounds
: > gpurun_out/ab.log
for round in 1 2; do
  for v in default "$@"; do
    if [ $v = default ]; then lib=$PWD/cornerstone-octree_amd/lib/libcstone_hip.so; else lib=$PWD/cornerstone-octree_amd/lib/variants/$v.so; fi
    for args in "" "--sorted"; do
      echo "== $v $args" >> gpurun_out/ab.log
      CSTONE_HIP_LIB=$lib timeout -k 10 100 python3 tools/sort_bench.py --reps 3 $args 2>&1 | grep "^pass:" >> gpurun_out/ab.log
    done
  done
done
cat gpurun_out/ab.log
