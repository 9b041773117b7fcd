#!/bin/bash
# A/B of the switches of a steady-state sync on one box (gpurun): every line is one bench.py / mr_bench.py run
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-plummer --no-cpu-baseline --no-mr-extra --no-variants --neighbor-targets 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms_per_step'].items() if v>0.005})"; }
mr() { echo "== mr $*"; env "$@" python tools/mr_bench.py --rccl --particles ${MRN:-1.25e7} --syncs 20 2>&1 | grep "ms per sync"; }
for rep in 1 2; do
run A=0
run CSTONE_DEVICE_GLOBAL_STEP=1
run CSTONE_NO_GATHER_OVERLAP=1
run CSTONE_DEVICE_GLOBAL_STEP=1 CSTONE_NO_GATHER_OVERLAP=1
done
mr A=0
mr CSTONE_SCAN_3PASS=1
