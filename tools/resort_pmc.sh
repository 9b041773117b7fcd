# PMC counters of leafSortKernel's counting path (static particles, CSTONE_RESORT_COUNT=1), one group per run
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/resort_pmc
rm -rf $O; mkdir -p $O
export CSTONE_RESORT_COUNT=${1:-1}
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  n=$(echo $c | tr " " "_" | cut -c1-24)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$n -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-plummer --neighbor-targets 0 --steps 3 --no-variants > /dev/null 2>&1
  echo "pmc $n done"
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "leafSortKernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
