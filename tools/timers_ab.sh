# what the stage timers (two HIP event records per bracket) cost a steady-state sync
R=$GRAFT_REPO_ROOT
for f in "" "--no-stage-timers" "" "--no-stage-timers"; do
  echo "== bench.py $f"
  timeout -k 10 300 python $R/bench.py --no-cpu-baseline --no-plummer --neighbor-targets 0 --steps 20 --warmup 3 --no-variants $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
