#!/usr/bin/env python3
"""condense the rocprofv3 output of tools/profile_r1.sh (gpurun_out/prof_<tag>/) into the small files kept under profiles/:
     <tag>_bench_kernel_stats.csv  -- rocprofv3 --kernel-trace --stats summary of `python bench.py`
     <tag>_sort_pmc_means.json     -- per-kernel means of the PMC counters of tools/sort_bench.py (1e8 pairs)
     <tag>_onesweep_traffic.json   -- HBM bytes per launch of the pass kernel: 2 x FETCH_SIZE (gfx950 reports half of a
                                      streamed read, MI355X_MICROARCH.md) + WRITE_SIZE, both in KiB units x 1024"""
import csv
import re
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
# per-kernel HBM traffic and achieved bandwidth of the sync's own kernels (steady-state launches over the full
# particle set: the largest grid of each kernel), from the two bench.py PMC runs
perk = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(src, f"benchpmc_{c}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "cship" not in name:
                continue
            m = re.search(r"(\w+Kernel)\b", name)
            if not m:
                continue
            short = m.group(1)
            if short == "leafSortKernel":
                # two instantiations share every grid: the one of the quiet tiles and the one of the tiles with movers
                short += "/moved" if re.search(r"true>\(", name) or name.rstrip().endswith("true>") or ", true>" in name else "/quiet"
            key = (short, int(row["Grid_Size"]))
            perk[key][c].append(float(row["Counter_Value"]))
            perk[key]["ns"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
if perk:
    # the full-size launches of a kernel: grids within 10 % of its largest one (the grids of the leaf kernels follow the
    # number of leaves, which differs a little from sync to sync and from run to run)
    top = defaultdict(int)
    for (short, grid) in perk:
        top[short] = max(top[short], grid)
    biggest = {}
    for (short, grid), v in perk.items():
        if grid < 0.9 * top[short]:
            continue
        g, acc = biggest.setdefault(short, (top[short], defaultdict(list)))
        for k, vals in v.items():
            acc[k].extend(vals)
    table = []
    for short, (grid, v) in sorted(biggest.items()):
        if not v.get("FETCH_SIZE") or not v.get("WRITE_SIZE"):
            continue
        # (a leaf-pass instantiation without tiles of its kind has nothing to do and moves no bytes: of these two kernels
        #  only the launches within a factor two of the largest count are averaged)
        idle = short.startswith("leafSortKernel")
        fs = [x for x in v["FETCH_SIZE"] if not idle or x >= 0.5 * max(v["FETCH_SIZE"])]
        ws = [x for x in v["WRITE_SIZE"] if not idle or x >= 0.5 * max(v["WRITE_SIZE"])]
        fetch = 2.0 * 1024 * sum(fs) / len(fs)
        write = 1024 * sum(ws) / len(ws)
        us = sum(v["ns"]) / len(v["ns"]) / 1e3
        table.append({"kernel": short, "grid_threads": grid, "launches": len(v["ns"]) // 2, "avg_us_under_pmc": us,
                      "hbm_read_bytes": fetch, "hbm_write_bytes": write,
                      "hbm_GBps": (fetch + write) / (us * 1e-6) / 1e9 if us > 0 else None})
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `python bench.py --steps 3` (N = 1e8), the full-size "
                         "launches of every kernel; FETCH_SIZE doubled (gfx950); durations are those of the counter runs",
               "kernels": table}, open(f"profiles/{tag}_kernel_hbm_traffic.json", "w"), indent=1)
# the multi-rank sync at the per-GPU size of the 8-GPU point (tools/mr_bench.py --rccl --particles 1.25e7): kernels per
# sync, averaged over the last 10 syncs of the trace (a sync starts with its one encodeHistogramKernel launch)
for trace in glob.glob(os.path.join(src, "mr", "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "encodeHistogramKernel" in r["Kernel_Name"]]
    if len(starts) < 12:
        continue
    use = starts[-11:]
    per = defaultdict(lambda: [0, 0.0])
    wall = 0.0
    for a, b in zip(use[:-1], use[1:]):
        wall += (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6
        for r in rows[a:b]:
            name = re.sub(r"^void |cship::\(anonymous namespace\)::|cship::", "", r["Kernel_Name"]).split("(")[0]
            if "at::native" in name or "elementwise" in name or "Cijk" in name:
                name = "torch kernels of the particle displacement (bench jiggle)"
            per[name][0] += 1
            per[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = len(use) - 1
    kernels = sorted(({"name": k, "launches_per_sync": v[0] / n, "us_per_sync": round(v[1] / n, 2)} for k, v in per.items()),
                     key=lambda e: -e["us_per_sync"])
    json.dump({"source": "rocprofv3 --kernel-trace of tools/mr_bench.py --rccl --particles 1.25e7 (cstone_hip_domain_mr_sync, "
                         "one rank, RCCL collectives from inside the library, 1 % of the particles displaced before every "
                         "sync), averages over the last 10 syncs",
               "launches_per_sync": sum(e["launches_per_sync"] for e in kernels),
               "kernel_time_ms_per_sync": sum(e["us_per_sync"] for e in kernels) / 1e3,
               "wall_ms_per_sync_under_profiler": wall / n, "kernels": kernels},
              open(f"profiles/{tag}_mr_sync_kernels.json", "w"), indent=1)
print("wrote", sorted(os.listdir("profiles")))

# the JSON line bench.py printed under the profiler (its live roofline number belongs next to the kernel stats)
blog = os.path.join(src, "bench_stdout.log") if "src" in dir() else "gpurun_out/prof_r1/bench_stdout.log"
if os.path.exists(blog):
    lines = [ln for ln in open(blog) if ln.startswith("{")]
    if lines:
        json.dump(json.loads(lines[0]), open(f"profiles/{tag}_bench_line_under_rocprof.json", "w"), indent=1)
