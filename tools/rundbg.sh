cd $GRAFT_REPO_ROOT
# debug / A-B of sort variants on one box: tools/rundbg.sh  (writes gpurun_out/dbg.log)
: > gpurun_out/dbg.log
V=$PWD/cornerstone-octree_amd/lib/variants
for v in dbglds; do
  echo "== $v 1e7" >> gpurun_out/dbg.log
  CSTONE_HIP_LIB=$V/$v.so timeout -k 10 60 python3 tools/sort_bench.py --n 1e7 --reps 1 2>&1 | grep -v amdgpu.ids | head -40 >> gpurun_out/dbg.log
done
for round in 1 2 3; do
  for v in early0 default; do
    if [ $v = default ]; then lib=$PWD/cornerstone-octree_amd/lib/libcstone_hip.so; else lib=$V/$v.so; fi
    for args in "" "--sorted"; do
      echo "== $v $args" >> gpurun_out/dbg.log
      CSTONE_HIP_LIB=$lib timeout -k 10 60 python3 tools/sort_bench.py --reps 3 $args 2>&1 | grep "^pass:\|Error\|error" >> gpurun_out/dbg.log
    done
  done
done
cat gpurun_out/dbg.log
