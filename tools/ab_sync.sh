#!/bin/bash
# A/B of the switches of a steady-state sync on one box (gpurun): every line is one bench.py run at 1e8 particles
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-plummer --no-cpu-baseline --no-mr-extra --no-variants --neighbor-targets 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms_per_step'].items() if v>0.005})"; }
run A=0
run CSTONE_NO_GATHER_OVERLAP=1
run CSTONE_SCAN_3PASS=1
run CSTONE_NO_GATHER_OVERLAP=1 CSTONE_SCAN_3PASS=1
run CSTONE_BENCH_SCRATCH=3
