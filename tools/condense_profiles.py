#!/usr/bin/env python3
"""condense the rocprofv3 output of tools/profile_r1.sh (gpurun_out/prof_<tag>/) into the small files kept under profiles/:
     <tag>_bench_kernel_stats.csv  -- rocprofv3 --kernel-trace --stats summary of `python bench.py`
     <tag>_sort_pmc_means.json     -- per-kernel means of the PMC counters of tools/sort_bench.py (1e8 pairs)
     <tag>_onesweep_traffic.json   -- HBM bytes per launch of the pass kernel: 2 x FETCH_SIZE (gfx950 reports half of a
                                      streamed read, MI355X_MICROARCH.md) + WRITE_SIZE, both in KiB units x 1024"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof_r1"
os.makedirs("profiles", exist_ok=True)
stats = glob.glob(os.path.join(src, "bench", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_bench_kernel_stats.csv")
# the pass kernel also runs on the small sorts of the linked-octree build: the --stats average mixes both, so the
# launches over the full particle set (grid = N / 16384 workgroups of 1024) are averaged separately from the trace
for trace in glob.glob(os.path.join(src, "bench", "**", "*kernel_trace.csv"), recursive=True):
    groups = defaultdict(list)
    for row in csv.DictReader(open(trace)):
        if "onesweepKernel<" in row["Kernel_Name"]:
            wgs = int(row["Grid_Size_X"]) // int(row["Workgroup_Size_X"])
            groups[wgs].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    summary = [{"workgroups": g, "launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3,
                "max_us": max(v) / 1e3} for g, v in sorted(groups.items())]
    json.dump({"kernel": "onesweepKernel", "source": "rocprofv3 --kernel-trace of `python bench.py --steps 5 --warmup 1`",
               "by_grid": summary}, open(f"profiles/{tag}_onesweep_launches.json", "w"), indent=1)
means = defaultdict(dict)
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        short = next((k for k in ("onesweepKernel", "onesweepTailKernel", "histogramKernel", "scanHistogramKernel")
                      if k + "<" in name or name.endswith(k)), None)
        if short is None or (short == "onesweepKernel" and "1024" not in name):
            continue
        a = acc[(short, row["Counter_Name"])]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
    for (k, c), (s, n) in acc.items():
        means[k][c] = s / n
json.dump(means, open(f"profiles/{tag}_sort_pmc_means.json", "w"), indent=1)
o = means.get("onesweepKernel", {})
if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
    fetch, write = 2.0 * o["FETCH_SIZE"] * 1024, o["WRITE_SIZE"] * 1024
    json.dump({"kernel": "onesweepKernel<u64,1024>", "n_pairs": 1e8, "fetch_bytes_corrected": fetch, "write_bytes": write,
               "traffic_bytes_per_launch": fetch + write,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/sort_bench.py --n 1e8; "
                       "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the streamed bytes)"},
              open(f"profiles/{tag}_onesweep_traffic.json", "w"), indent=1)
print("wrote", sorted(os.listdir("profiles")))
