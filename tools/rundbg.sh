cd $GRAFT_REPO_ROOT
# A-B of sort variants on one box: tools/rundbg.sh v1 v2 ...  (writes gpurun_out/dbg.log); "default" = the regular build
: > gpurun_out/dbg.log
V=$PWD/cornerstone-octree_amd/lib/variants
for round in 1 2 3; do
  for v in "$@"; do
    if [ $v = default ]; then lib=$PWD/cornerstone-octree_amd/lib/libcstone_hip.so; else lib=$V/$v.so; fi
    for args in "" "--sorted"; do
      echo "== $v $args" >> gpurun_out/dbg.log
      CSTONE_HIP_LIB=$lib timeout -k 10 60 python3 tools/sort_bench.py --reps 3 $args 2>&1 | grep "^pass:\|Error\|error" >> gpurun_out/dbg.log
    done
  done
done
cat gpurun_out/dbg.log
