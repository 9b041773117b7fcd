#!/usr/bin/env python3
"""stage timers of Domain::sync when 1 % of the particles move between the syncs (bench.py's extras.moving_particles)"""
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
for moving in (False, True):
    for _ in range(2):
        if moving:
            pipe.jiggle()
        pipe.step()
    ctx.profile_enable(True)
    ctx.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 8
    for _ in range(steps):
        if moving:
            pipe.jiggle()
        pipe.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = {s: round(ctx.profile_get(s)[0] / steps, 3) for s in cstone_amd.STAGES}
    ln = {s: ctx.profile_get(s)[1] // steps for s in cstone_amd.STAGES}
    print(("moving" if moving else "static"), f"{dt*1e3:.2f} ms/sync", st, "launches", ln, flush=True)
    ctx.profile_enable(False)
