cd $GRAFT_REPO_ROOT
# usage: runvar.sh [variant ...]   -- parity, sort bench of the default build, then of every named variant, trace report
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
: > gpurun_out/variants.log
for args in "" "--sorted" "--key-bits 32"; do
  echo "== default $args" >> gpurun_out/variants.log
  timeout -k 10 100 python3 tools/sort_bench.py --reps 3 $args 2>&1 | tail -1 >> gpurun_out/variants.log
done
for v in "$@"; do
  echo "== $v" >> gpurun_out/variants.log
  CSTONE_HIP_LIB=$PWD/cornerstone-octree_amd/lib/variants/$v.so timeout -k 10 100 python3 tools/sort_bench.py --reps 3 2>&1 | tail -1 >> gpurun_out/variants.log
done
cat gpurun_out/variants.log
if [ -f cornerstone-octree_amd/lib/variants/trace.so ]; then
  CSTONE_HIP_LIB=$PWD/cornerstone-octree_amd/lib/variants/trace.so CSTONE_SORT_TRACE_FILE=$PWD/gpurun_out/trace.bin timeout -k 10 200 python tools/sort_bench.py --reps 1 > gpurun_out/trace_run.log 2>&1
  python tools/sort_trace.py gpurun_out/trace.bin > gpurun_out/trace_report.txt; rm -f gpurun_out/trace.bin; head -4 gpurun_out/trace_report.txt
fi
