# workgroups per CU of the grid-stride encode kernels (CSTONE_ENCODE_BLOCKS), steady-state sync
R=$GRAFT_REPO_ROOT
for b in 4 6 8 10 12 16; do
  echo -n "blocks/CU $b: "
  CSTONE_ENCODE_BLOCKS=$b timeout -k 10 300 python $R/bench.py --no-cpu-baseline --no-plummer --neighbor-targets 0 --steps 10 --warmup 2 --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['stage_ms_per_step']['encode'],4))"
done
