# rocprofv3 runs whose summaries are copied into profiles/ (run through gpurun)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r1
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/bench -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plummer > $O/bench_stdout.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr " " "_" | cut -c1-24)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$n -o p --output-format csv -- python3 $R/tools/sort_bench.py --reps 1 > /dev/null 2>&1
done
# HBM traffic of every kernel of the sync itself (bench.py, 3 timed syncs), again one counter per run
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $O/benchpmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-plummer --neighbor-targets 0 > /dev/null 2>&1
done
ls $O
