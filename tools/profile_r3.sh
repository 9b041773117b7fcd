# rocprofv3 runs whose summaries are copied into profiles/ (run through gpurun): tools/profile_r3.sh, then
# python tools/condense_profiles.py r03 gpurun_out/prof_r3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r3
rm -rf $O; mkdir -p $O
# the bench's own command under the kernel trace (its JSON line belongs next to the kernel stats)
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/bench -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plummer > $O/bench_stdout.log 2>&1
echo "bench trace done" > $O/progress.log
# HBM traffic of every kernel of the timed loop (bench.py, 3 timed syncs with drifting particles), one counter per run
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $O/benchpmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-plummer --no-variants --neighbor-targets 0 > /dev/null 2>&1
  echo "bench pmc $c done" >> $O/progress.log
done
# the multi-rank sync at the per-GPU size of the 8-GPU strong-scaling point, RCCL world of one rank
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/mr -o mr --output-format csv -- python3 $R/tools/mr_bench.py --rccl --particles 1.25e7 --syncs 14 > $O/mr_stdout.log 2>&1
echo "mr trace done" >> $O/progress.log
ls $O
# what the counters say for the gathers' access pattern on known byte counts (tools/gather_calib.py)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/gather_calib_$c -o p --output-format csv -- python3 $R/tools/gather_calib.py > $O/gather_calib_$c.log 2>&1
done
echo "gather calibration done" >> $O/progress.log
# the same multi-rank sync as a sequence of HIP API calls with their kernels (tools/mr_trace.py)
timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace -d $O/mr_api -o mr --output-format csv -- python3 $R/tools/mr_bench.py --rccl --particles 1.25e7 --syncs 8 > $O/mr_api_stdout.log 2>&1
echo "mr api trace done" >> $O/progress.log
