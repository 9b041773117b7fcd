cd $GRAFT_REPO_ROOT
CSTONE_HIP_LIB=$PWD/cornerstone-octree_amd/lib/variants/trace.so CSTONE_SORT_TRACE_FILE=$PWD/gpurun_out/trace.bin timeout -k 10 200 python tools/sort_bench.py --reps 1 > gpurun_out/trace_run.log 2>&1
python tools/sort_trace.py gpurun_out/trace.bin > gpurun_out/trace_report.txt; rm -f gpurun_out/trace.bin; head -4 gpurun_out/trace_report.txt
