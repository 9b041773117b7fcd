#!/usr/bin/env python3
"""long random walk of the re-sorting domain against the never-re-sorting one (the loop of
tests/test_resort.py::test_resort_random_walk, more steps, more particles): python tools/resort_soak.py [seeds] [steps]
[gentle] -- gentle: only moves that leave the re-sort in play (jitter, jumps, equal keys, removals)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cstone_amd  # noqa: E402
from test_resort import _Stepper  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
hip = cstone_amd.Context(0)
t0 = time.time()
for seed in range(seeds):
    rng = np.random.default_rng(7000 + seed)
    kb = int(rng.choice([32, 64]))
    bucket_focus = int(rng.choice([8, 32, 64, 64, 150, 256]))
    bc = tuple(int(v) for v in rng.choice([0, 1], 3))
    n = int(rng.choice([50_000, 300_000, 1_500_000]))
    a_, b_ = (_Stepper(hip, kb, 64, bucket_focus, 1, bc, n, 77 + seed, allow) for allow in (True, False))
    kinds = ["none", "jitter", "jitter", "jitter", "few", "few", "pairs", "remove", "shuffle", "many", "collapse"]
    if len(sys.argv) > 3 and sys.argv[3] == "gentle":
        kinds = ["none", "jitter", "jitter", "few", "few", "pairs", "remove"]
    for step in range(steps):
        kind = "none" if step == 0 else str(rng.choice(kinds))
        if step:
            a_.move(kind, np.random.default_rng(10_000 * seed + step))
            b_.move(kind, np.random.default_rng(10_000 * seed + step))
        a, b = a_.sync(), b_.sync()
        va, vb = a["view"], b["view"]
        ok = (va.end_index, va.num_focus_leaves) == (vb.end_index, vb.num_focus_leaves)
        for f in ("keys", "x", "y", "z", "h", "ident"):
            ok = ok and np.array_equal(a[f], b[f])
        m, L = va.end_index, va.num_focus_leaves
        ok = ok and np.array_equal(a_.dom.fetch(va.sfc_order, m, np.uint32), b_.dom.fetch(vb.sfc_order, m, np.uint32))
        ok = ok and np.array_equal(a_.dom.fetch(va.layout, L + 1, np.uint32), b_.dom.fetch(vb.layout, L + 1, np.uint32))
        ok = ok and np.array_equal(a_.dom.fetch(va.focus_leaf_counts, L, np.uint32),
                                   b_.dom.fetch(vb.focus_leaf_counts, L, np.uint32))
        if not ok:
            print(f"MISMATCH seed {seed} step {step} ({kind}) kb {kb} bucket {bucket_focus} n {n} bc {bc}", flush=True)
            sys.exit(1)
    print(f"seed {seed}: kb {kb} bucket {bucket_focus} n {n} bc {bc}: {steps} steps equal, {a_.dom.stats()} "
          f"[{time.time() - t0:.0f} s]", flush=True)
print("soak ok")
