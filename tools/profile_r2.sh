# rocprofv3 runs whose summaries are copied into profiles/ (run through gpurun): tools/profile_r2.sh, then
# python tools/condense_profiles.py r02 gpurun_out/prof_r2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r2
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/bench -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plummer > $O/bench_stdout.log 2>&1
echo "bench trace done" > $O/progress.log
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  n=$(echo $c | tr " " "_" | cut -c1-24)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$n -o p --output-format csv -- python3 $R/tools/sort_bench.py --reps 1 > /dev/null 2>&1
  echo "pmc $n done" >> $O/progress.log
done
# HBM traffic of every kernel of the sync itself (bench.py, 3 timed syncs), again one counter per run
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $O/benchpmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-plummer --neighbor-targets 0 > /dev/null 2>&1
  echo "bench pmc $c done" >> $O/progress.log
done
# the multi-rank sync at the per-GPU size of the 8-GPU strong-scaling point, RCCL world of one rank
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/mr -o mr --output-format csv -- python3 $R/tools/mr_bench.py --rccl --particles 1.25e7 --syncs 14 > $O/mr_stdout.log 2>&1
echo "mr trace done" >> $O/progress.log
ls $O
