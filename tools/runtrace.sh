cd $GRAFT_REPO_ROOT
# phase budget of the pass kernel (CSTONE_SORT_TRACE build) for random and for sorted keys
L=$PWD/cornerstone-octree_amd/lib/variants/trace.so
for args in "" "--sorted"; do
  CSTONE_HIP_LIB=$L CSTONE_SORT_TRACE_FILE=$PWD/gpurun_out/trace.bin timeout -k 10 200 python tools/sort_bench.py --reps 1 $args > gpurun_out/trace_run.log 2>&1
  echo "== trace $args" >> gpurun_out/trace_report.txt
  python tools/sort_trace.py gpurun_out/trace.bin >> gpurun_out/trace_report.txt; rm -f gpurun_out/trace.bin
done
grep -A3 "== trace\|^pass [07]" gpurun_out/trace_report.txt | head -60
