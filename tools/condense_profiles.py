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
from statistics import median

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
# every kernel of the library: the --stats average mixes a kernel's launches over the full particle set with its small
# ones (tree-sized sorts, launches of the leaf pass that find no tile of their kind); here the launches whose grid is
# within 10 % of the kernel's largest one are averaged on their own -- the number bench.py's live stage timers must match
for trace in glob.glob(os.path.join(src, "bench", "**", "*kernel_trace.csv"), recursive=True):
    groups = defaultdict(list)
    for row in csv.DictReader(open(trace)):
        name = row["Kernel_Name"]
        if "cship" not in name:
            continue
        m = re.search(r"(\w+Kernel)\b", name)
        if not m:
            continue
        short = m.group(1)
        if short == "leafSortKernel":
            short += "/counting" if ", true>" in name else "/quiet"
        groups[short].append((int(row["Grid_Size_X"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    out = []
    for short, v in sorted(groups.items()):
        top = max(g for g, _ in v)
        full = [ns for g, ns in v if g >= 0.9 * top]
        if short.startswith("leafSortKernel"):  # launches that find no tile of their kind return at once
            full = [ns for ns in full if ns >= 0.5 * max(full)]
        out.append({"kernel": short, "grid_threads": top, "launches_total": len(v), "launches_full_size": len(full),
                    "avg_us_full_size": sum(full) / len(full) / 1e3, "median_us_full_size": median(full) / 1e3, "min_us": min(full) / 1e3, "max_us": max(full) / 1e3,
                    "total_ms_all_launches": sum(ns for _, ns in v) / 1e6})
    out.sort(key=lambda e: -e["total_ms_all_launches"])
    json.dump({"source": "rocprofv3 --kernel-trace of `python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plummer` "
                         "(headline loop and the extras' loops), launches over the full particle set of every kernel",
               "kernels": out}, open(f"profiles/{tag}_bench_kernel_launches.json", "w"), indent=1)
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
if means:  # (only when the sort's PMC passes were part of the run: no empty files)
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
                short += "/counting" if ", true>" in name else "/quiet"
            key = (short, int(row["Grid_Size"]))
            perk[key][c].append(float(row["Counter_Value"]))
            perk[key]["ns"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
# tools/gather_calib.py under the same counters: three launches each through the identity, a permutation inside blocks of
# 64 elements, a random permutation of everything
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(src, f"gather_calib_{c}", "**", "*counter_collection.csv"), recursive=True):
        rows = sorted((r for r in csv.DictReader(open(f)) if "gatherMultiKernel" in r["Kernel_Name"]),
                      key=lambda r: int(r["Start_Timestamp"]))
        for i, name in enumerate(("identity", "inside_blocks_of_64", "random")):
            part = rows[3 * i:3 * i + 3]
            if part:
                e = calib.setdefault(name, {"elements": 1e8, "algorithmic_read_bytes": 28e8, "algorithmic_write_bytes": 24e8})
                e[c + "_raw_bytes"] = 1024 * median([float(r["Counter_Value"]) for r in part])
                e["us_" + c] = median([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in part]) / 1e3
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
        # medians: the first sync of a run sorts a random cloud from scratch and its gathers follow a random permutation
        # (14 x the fetched bytes of a steady-state launch, tools/gather_calib.py); the steady state is what is reported
        fetch = 2.0 * 1024 * median(fs)
        write = 1024 * median(ws)
        us = median(v["ns"]) / 1e3
        table.append({"kernel": short, "grid_threads": grid, "launches": len(v["ns"]) // 2, "median_us_under_pmc": us,
                      "hbm_read_bytes": fetch, "hbm_write_bytes": write,
                      "hbm_GBps": (fetch + write) / (us * 1e-6) / 1e9 if us > 0 else None})
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `python bench.py --steps 3` (N = 1e8), medians over "
                         "the full-size launches of every kernel; FETCH_SIZE doubled (gfx950: checked for the gathers' "
                         "8-byte accesses on known byte counts, gather_calibration below); durations are those of the "
                         "counter runs",
               "particles": 1e8, "gather_calibration": calib,
               "kernels": table}, open(f"profiles/{tag}_kernel_hbm_traffic.json", "w"), indent=1)
# the multi-rank sync at the per-GPU size of the 8-GPU point (tools/mr_bench.py --rccl --particles 1.25e7): kernels per
# sync, averaged over the last 10 syncs of the trace (a sync is counted from its encode launch to the next one's: the
# leaf-table kernels in front of the encode belong to the following sync in this bookkeeping, the sums are unaffected)
for trace in glob.glob(os.path.join(src, "mr", "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    # (a sync computes its keys exactly once: in encodeResortKernel when it re-sorts, in encodeHistogramKernel otherwise)
    starts = [i for i, r in enumerate(rows) if "encodeHistogramKernel" in r["Kernel_Name"] or "encodeResortKernel" in r["Kernel_Name"]]
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
                name = "torch kernels of the particle displacement (between the syncs, not part of them)"
            per[name][0] += 1
            per[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = len(use) - 1
    kernels = sorted(({"name": k, "launches_per_sync": v[0] / n, "us_per_sync": round(v[1] / n, 2)} for k, v in per.items()),
                     key=lambda e: -e["us_per_sync"])
    json.dump({"source": "rocprofv3 --kernel-trace of tools/mr_bench.py --rccl --particles 1.25e7 (cstone_hip_domain_mr_sync, "
                         "one rank, RCCL collectives from inside the library, every particle displaced by <= 0.1 h before "
                         "every sync), averages over the last 10 syncs",
               "launches_per_sync": sum(e["launches_per_sync"] for e in kernels if not e["name"].startswith("torch kernels")),
               "kernel_time_ms_per_sync": sum(e["us_per_sync"] for e in kernels if not e["name"].startswith("torch kernels")) / 1e3,
               "note": "the sums leave out the torch kernels of the particle displacement between the syncs; the wall time is from the start of one sync to the start of the next, displacement included",
               "wall_ms_per_sync_under_profiler": wall / n, "kernels": kernels},
              open(f"profiles/{tag}_mr_sync_kernels.json", "w"), indent=1)
if glob.glob(os.path.join(src, "mr_api", "*hip_api_trace.csv")):
    import subprocess
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "mr_trace.py"),
                    os.path.join(src, "mr_api"), "--json", f"profiles/{tag}_mr_sync_api_sequence.json"],
                   check=True, stdout=subprocess.DEVNULL)
# wall time per multi-rank sync without a profiler (tools/mr_bench.py --rccl, world of one rank)
plain = os.path.join(src, "mr_plain.log")
if os.path.exists(plain):
    import re
    rows = []
    for line in open(plain):
        m = re.search(r"(\d+) RCCL rank\(s\) on one GPU, ([\d.e+]+) particles: ([\d.]+) ms per sync", line)
        if m:
            rows.append({"rccl_ranks": int(m.group(1)), "particles": float(m.group(2)), "ms_per_sync": float(m.group(3))})
    json.dump({"source": "python3 tools/mr_bench.py --rccl --particles N --syncs 20 (no profiler): cstone_hip_domain_mr_sync, RCCL "
                         "world of one rank, every particle displaced by <= 0.1 h before every sync (outside the timed "
                         "intervals), wall clock around each sync", "runs": rows},
              open(f"profiles/{tag}_mr_sync_times.json", "w"), indent=1)

# roctx ranges of the stages (cstone_hip_profile_markers; rocprofv3 --marker-trace): time per stage and sync from the
# marker trace alone, without the event brackets of cstone_hip_profile_enable
for trace in glob.glob(os.path.join(src, "markers", "**", "*marker_api_trace.csv"), recursive=True):
    per = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(trace)):
        name = row.get("Function") or row.get("Name") or ""
        if not name.startswith("cstone:"):
            continue
        per[name][0] += 1
        per[name][1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    if per:
        json.dump({"source": "rocprofv3 --marker-trace --kernel-trace of `CSTONE_BENCH_MARKERS=1 python bench.py --steps 3 ...`: "
                             "roctx ranges cstone:<stage> pushed by the library's stage timers (HOST time between push and "
                             "pop: the launches of a stage, not the kernels' run time)",
                   "ranges": sorted(({"range": k, "count": v[0], "host_us_total": round(v[1], 1)} for k, v in per.items()),
                                    key=lambda e: -e["host_us_total"])},
                  open(f"profiles/{tag}_stage_marker_ranges.json", "w"), indent=1)
# the numbers README.md and DESIGN.md quote, in one place and straight from the files above (and from
# profiles/{tag}_bench_line.json, the default `python bench.py` run of the same tree without a profiler)
def _numbers():
    out = [f"# {tag}: the numbers the documents quote (generated by tools/condense_profiles.py from profiles/{tag}_*)", ""]
    path = f"profiles/{tag}_bench_line.json"
    if os.path.exists(path):
        b = json.load(open(path))
        ex = b.get("extras", {})
        out += [f"* headline (`{tag}_bench_line.json`): {b['ms_per_step']:.3f} ms per sync = {b['value']:.3e} {b['unit']}, "
                f"n_gpus {b['n_gpus']}, first sync {b.get('first_sync_ms', 0):.1f} ms"]
        for k in b["roofline"]["kernels"]:
            alone = k.get("alone")
            out += [f"  * `{k['kernel']}`: {k['bytes_per_launch'] / 1e9:.2f} GB in {k['avg_ms']:.3f} ms = {k['achieved'] / 1e3:.2f} "
                    f"TB/s = {k['frac']:.3f} of the peak; PMC traffic {((k['traffic'] or 0) / 1e9):.2f} GB"
                    + (f"; alone {alone['avg_ms']:.3f} ms = {alone['frac']:.3f}" if alone else "")]
        sy, ow = b["roofline"]["sync"], b["roofline"].get("onesweep") or {}
        out += [f"  * whole sync: {sy['algorithmic_bytes_per_step'] / 1e9:.1f} GB per sync = {sy['achieved'] / 1e3:.2f} TB/s = "
                f"{sy['frac']:.3f}"]
        if ow:
            out += [f"  * digit pass: {ow['avg_launch_ms']:.4f} ms per launch = {ow['frac']:.3f} of the peak "
                    f"({ow['launches']} launches, {ow['measured_on']})"]
            ws = ow.get("whole_sort")
            if ws:
                out += [f"  * whole sort on the {ws['model_bytes_per_pair']} B/pair model: {ws['ms']:.3f} ms = "
                        f"{ws['achieved'] / 1e3:.2f} TB/s = {ws['frac']:.3f}"]
        out += ["  * stages (ms per sync): " + ", ".join(f"{k} {v:.3f}" for k, v in b["stage_ms_per_step"].items() if v)]
        for name in ("one_stream", "moving_particles", "zero_motion", "sorted_from_scratch", "all_digits_sorted",
                     "open_box_moving_extremes", "encode_sort_tree_1e7", "mr_path_world_of_one"):
            if name in ex and isinstance(ex[name], dict) and "ms_per_step" in ex[name]:
                out += [f"* extras.{name}: {ex[name]['ms_per_step']:.3f} ms"]
        if "plummer" in ex:
            pl = ex["plummer"]
            out += [f"* extras.plummer: {pl['ms_per_step']:.3f} ms drifting ({pl['syncs']}), "
                    f"{pl['zero_motion']['ms_per_step']:.3f} ms with nothing moving, {pl['focus_leaves']} leaves, "
                    f"findNeighbors {pl['find_neighbors']['targets_per_s']:.3e} targets/s at "
                    f"{pl['find_neighbors']['mean_neighbors']:.1f} neighbours"]
        fn = ex.get("find_neighbors")
        if fn:
            out += [f"* extras.find_neighbors: {fn[0]['targets_per_s']:.3e} targets/s, {fn[0]['mean_neighbors']:.1f} neighbours"]
        cb = b.get("cpu_baseline")
        if cb:
            out += [f"* cpu_baseline ({cb['kind']}, {cb['cores']} cores): {cb['value']:.3e} {cb['unit']}"]
    path = f"profiles/{tag}_mr_sync_times.json"
    if os.path.exists(path):
        out += ["* multi-rank sync, RCCL world of one, no profiler (`%s_mr_sync_times.json`): " % tag +
                ", ".join(f"{r['particles']:.3g}: {r['ms_per_sync']} ms" for r in json.load(open(path))["runs"])]
    path = f"profiles/{tag}_mr_sync_api_sequence.json"
    if os.path.exists(path):
        a = json.load(open(path))
        out += [f"* multi-rank sync at 1.25e7, API calls per sync (`{tag}_mr_sync_api_sequence.json`): {a['api_calls_per_sync']}, "
                f"kernels and copies on the GPU {a['gpu_busy_us']:.0f} us"]
    path = f"profiles/{tag}_bench_kernel_launches.json"
    if os.path.exists(path):
        rows = {r["kernel"]: r for r in json.load(open(path))["kernels"]}
        for k in ("onesweepKernel", "encodeResortKernel", "leafSortWaveKernel", "gatherMultiKernel", "gatherHaloRadiiKernel",
                  "binMoversKernel", "placeMoversKernel"):
            if k in rows:
                r = rows[k]
                out += [f"* rocprofv3 kernel trace, `{k}`: {r['launches_full_size']} full-size launches, average "
                        f"{r['avg_us_full_size']:.1f} us, median {r['median_us_full_size']:.1f} us"]
    open(f"profiles/{tag}_numbers.md", "w").write("\n".join(out) + "\n")


_numbers()
print("wrote", sorted(os.listdir("profiles")))

# the JSON line bench.py printed under the profiler (its live roofline number belongs next to the kernel stats)
blog = os.path.join(src, "bench_stdout.log") if "src" in dir() else "gpurun_out/prof_r1/bench_stdout.log"
if os.path.exists(blog):
    lines = [ln for ln in open(blog) if ln.startswith("{")]
    if lines:
        json.dump(json.loads(lines[0]), open(f"profiles/{tag}_bench_line_under_rocprof.json", "w"), indent=1)
