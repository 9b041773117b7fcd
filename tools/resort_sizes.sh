# where the incremental re-sort starts to pay: steady-state sync with and without it, by particle count
R=$GRAFT_REPO_ROOT
for n in 3e5 1e6 3e6 1e7 3e7; do
  for f in "" 1; do
    if [ -n "$f" ]; then export CSTONE_NO_RESORT=1; else unset CSTONE_NO_RESORT; fi
    echo -n "n=$n no_resort=${f:-0}: "
    timeout -k 10 300 python $R/bench.py --particles $n --no-cpu-baseline --no-plummer --neighbor-targets 0 --steps 20 --warmup 3 --no-variants --no-stage-timers 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"
  done
done
