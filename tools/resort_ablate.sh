# timing of the leaf pass on static particles (quiet path / counting path forced), per build variant
# (tools/build_variant.sh NAME FLAGS resort)
R=$GRAFT_REPO_ROOT
for lib in "" $RESORT_VARIANTS; do
  for m in "" 1; do
    echo "== ${lib:-default} CSTONE_RESORT_COUNT=$m"
    if [ -n "$lib" ]; then export CSTONE_HIP_LIB=$R/cornerstone-octree_amd/lib/variants/$lib.so; else unset CSTONE_HIP_LIB; fi
    if [ -n "$m" ]; then export CSTONE_RESORT_COUNT=1; else unset CSTONE_RESORT_COUNT; fi; timeout -k 10 300 python $R/bench.py --no-cpu-baseline --no-plummer --neighbor-targets 0 --steps 5 --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms_per_step']['resort_leaves'])"
  done
done
