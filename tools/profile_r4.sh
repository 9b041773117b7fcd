# rocprofv3 runs whose summaries are copied into profiles/ (run through gpurun): tools/profile_r4.sh, then
# python tools/condense_profiles.py r04 gpurun_out/prof_r4
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r4
rm -rf $O; mkdir -p $O
# the bench's own command under the kernel trace (its JSON line belongs next to the kernel stats)
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/bench -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plummer --no-mr-extra > $O/bench_stdout.log 2>&1
echo "bench trace done" > $O/progress.log
# HBM traffic of every kernel of the timed loop (bench.py, 3 timed syncs with drifting particles), one counter per run
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $O/benchpmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-plummer --no-mr-extra --no-variants --neighbor-targets 0 > /dev/null 2>&1
  echo "bench pmc $c done" >> $O/progress.log
done
# the multi-rank sync at the per-GPU size of the 8-GPU strong-scaling point, RCCL world of one rank
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/mr -o mr --output-format csv -- python3 $R/tools/mr_bench.py --rccl --particles 1.25e7 --syncs 14 > $O/mr_stdout.log 2>&1
echo "mr trace done" >> $O/progress.log
# ... and without a profiler around it: the wall time per sync at four sizes (condensed into r04_mr_sync_times.json)
for n in 1.25e7 2.5e7 5e7 1e8; do
  timeout -k 10 200 python3 $R/tools/mr_bench.py --rccl --particles $n --syncs 20 2>&1 | grep "ms per sync" >> $O/mr_plain.log
done
echo "mr plain done" >> $O/progress.log
# the same multi-rank sync as a sequence of HIP API calls with their kernels (tools/mr_trace.py)
timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace -d $O/mr_api -o mr --output-format csv -- python3 $R/tools/mr_bench.py --rccl --particles 1.25e7 --syncs 8 > $O/mr_api_stdout.log 2>&1
echo "mr api trace done" >> $O/progress.log
# the radix sort on its own: 1e8 random 64-bit pairs (the digit pass the north star singles out)
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/sort -o sort --output-format csv -- python3 $R/tools/sort_bench.py > $O/sort_stdout.log 2>&1
echo "sort trace done" >> $O/progress.log
# roctx ranges of the stages (cstone_hip_profile_markers): the stage table of a sync from the marker trace
CSTONE_BENCH_MARKERS=1 timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace -d $O/markers -o m --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-plummer --no-mr-extra --no-variants --neighbor-targets 0 > $O/markers_stdout.log 2>&1
echo "marker trace done" >> $O/progress.log
ls $O
