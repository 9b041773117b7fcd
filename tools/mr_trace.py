#!/usr/bin/env python3
"""one steady-state cstone_hip_domain_mr_sync as a sequence of HIP API calls with the kernels they launched, from
`rocprofv3 --hip-trace --kernel-trace -d DIR -o mr --output-format csv -- python3 tools/mr_bench.py ...`:
usage: mr_trace.py DIR [--json OUT].  Prints launches, copies, memsets and stream synchronisations in order and their
counts per sync (the interval between two torch.cuda.synchronize() calls of the bench with the most launches)."""
import collections
import csv
import glob
import json
import re
import sys

d = sys.argv[1]
api = list(csv.DictReader(open(glob.glob(d + "/*hip_api_trace.csv")[0])))
ker = {r["Correlation_Id"]: r for r in csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0]))}
api.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Function"] for r in api]
ds = [i for i, n in enumerate(names) if n == "hipDeviceSynchronize"]
best = None
for a, b in zip(ds[:-1], ds[1:]):
    if sum(1 for r in api[a + 1:b] if r["Function"] == "hipLaunchKernel") > 40:
        best = (a, b)
a, b = best
t0 = int(api[a + 1]["Start_Timestamp"])
shown = ("hipLaunchKernel", "hipMemcpyAsync", "hipStreamSynchronize", "hipMemsetAsync", "hipMemcpy")
counts = collections.Counter()
seq = []
gpu_us = 0.0
for r in api[a + 1:b]:
    f = r["Function"]
    if f not in shown:
        continue
    counts[f] += 1
    what = ""
    if r["Correlation_Id"] in ker:
        k = ker[r["Correlation_Id"]]
        us = (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3
        gpu_us += us
        what = re.sub(r"^void |cship::\(anonymous namespace\)::|cship::", "", k["Kernel_Name"]).split("(")[0][:70]
        # (when the kernel ran on the GPU, relative to the first call of the sync, and on which HSA queue: kernels of the
        #  second stream overlap those of the first)
        what = (f"{what} [{us:.1f} us; gpu {(int(k['Start_Timestamp']) - t0) / 1e3:.1f}..{(int(k['End_Timestamp']) - t0) / 1e3:.1f}"
                f" q{k.get('Queue_Id', '?')}]")
    seq.append((round((int(r["Start_Timestamp"]) - t0) / 1e3, 1), f, what))
    print(f"{seq[-1][0]:8.1f} {f:22s} {what}")
wall = (int(api[b]["Start_Timestamp"]) - t0) / 1e3
print(dict(counts), f"wall under the tracer {wall:.0f} us, kernels and copies on the GPU {gpu_us:.0f} us")
if "--json" in sys.argv:
    json.dump({"source": "rocprofv3 --hip-trace --kernel-trace of tools/mr_bench.py --rccl --particles 1.25e7, one steady-state "
                         "sync (every particle displaced by <= 0.1 h before it)",
               "api_calls_per_sync": dict(counts), "wall_us_under_tracer": wall, "gpu_busy_us": gpu_us,
               "sequence": [{"t_us": t, "call": f, "kernel": w} for t, f, w in seq]},
              open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
