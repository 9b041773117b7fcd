#!/usr/bin/env python3
"""stage timers of the steady-state Domain::sync for three kinds of motion between the syncs -- none, 1 % of the particles
displaced by <= 2h (bench.py's extras.moving_particles), every particle displaced by <= 0.1 h (the headline workload) --
with the displacement OUTSIDE the timed interval.  usage: leafpass_ab.py [particles]; environment as for bench.py
(CSTONE_RESORT_SCAN=1: the round-2 formulation of the leaf pass for tiles with movers)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))
import torch  # noqa: E402

import cstone_amd  # noqa: E402
from bench import SyncPipeline  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ctx = cstone_amd.Context(0)
pipe = SyncPipeline(ctx, n, 64, 64, "hilbert", max(64, n // 100), 64, seed=42)
pipe.first_sync()
for mode, move in (("static", None), ("jiggle 1%", pipe.jiggle), ("drift 0.1h", pipe.drift)):
    for _ in range(2):
        if move:
            move()
        pipe.step()
    ctx.profile_enable(True)
    ctx.profile_reset()
    steps, total = 8, 0.0
    st0 = pipe.dom.stats()
    for _ in range(steps):
        if move:
            move()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.step()
        torch.cuda.synchronize()
        total += time.perf_counter() - t0
    st1 = pipe.dom.stats()
    st = {s: round(ctx.profile_get(s)[0] / steps, 3) for s in cstone_amd.STAGES if ctx.profile_get(s)[1]}
    print(f"{mode:12s} {total / steps * 1e3:6.2f} ms/sync (timers on)", st,
          {k: st1[k] - st0[k] for k in ("resorts", "resort_fallbacks", "box_redos")}, "movers", st1["last_movers"], flush=True)
    ctx.profile_enable(False)
