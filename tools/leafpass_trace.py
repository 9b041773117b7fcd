#!/usr/bin/env python3
"""phase budget of the leaf pass for tiles with movers (leafSortWaveKernel, the flavour with two elements per lane:
run with CSTONE_RESORT_PAIRS=1 at sizes where it is not the default) from a CSTONE_RESORT_TRACE build:
   tools/build_variant.sh rtrace -DCSTONE_RESORT_TRACE resort
   CSTONE_HIP_LIB=.../lib/variants/rtrace.so python tools/leafpass_trace.py [particles] [jiggle|drift]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cstone_amd  # noqa: E402
from bench import SyncPipeline  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "drift"
ctx = cstone_amd.Context(0)
pipe = SyncPipeline(ctx, n, 64, 64, "hilbert", max(64, n // 100), 64, seed=42)
pipe.first_sync()
move = pipe.drift if kind == "drift" else pipe.jiggle
for _ in range(3):
    move()
    pipe.step()
SLOTS, rows = 12, 1 << 17
buf = torch.zeros(rows * SLOTS, dtype=torch.int64, device="cuda")
assert ctx.lib.cstone_hip_resort_trace_set(C.c_void_p(buf.data_ptr())) == 0
move()
torch.cuda.synchronize()
pipe.step()
torch.cuda.synchronize()
ctx.lib.cstone_hip_resort_trace_set(C.c_void_p(0))
t = buf.cpu().numpy().reshape(rows, SLOTS)
t = t[(t[:, :3] > 0).all(axis=1)].astype(np.int64)
# leafSortWaveKernel, wave 0 of every workgroup: [0] start, [1] tables in LDS, [2] last step done, [8] stores drained (wall
# clock, 100 MHz); [3] steps, [4..7] shader-clock cycles spent issuing the next step's loads / waiting for this step's keys /
# in the network / placing and storing
us = lambda a, b: (t[:, b] - t[:, a]) * 10.0 / 1e3
span = (t[:, 8].max() - t[:, 0].min()) * 10.0 / 1e3
life = us(0, 8)
print(f"{kind}, {n:.1e} particles: {t.shape[0]} tiles traced, kernel span {span:.1f} us, tile lifetime mean {life.mean():.2f} us "
      f"(p50 {np.median(life):.2f}, p95 {np.percentile(life, 95):.2f}), mean tiles alive {life.sum() / span:.0f}")
print(f"   mean us: tables {us(0, 1).mean():.2f}   steps {us(1, 2).mean():.2f}   drain {us(2, 8).mean():.2f};   steps per wave "
      f"{t[:, 3].mean():.1f}")
cyc = t[:, 4:8].astype(np.float64)
tot = cyc.sum(axis=1)
print("   cycles per step (wave 0): " + "  ".join(f"{a} {v:.0f}" for a, v in zip(
    ["issue next loads", "wait for keys", "network", "place + store"], (cyc / t[:, 3:4]).mean(axis=0))) +
    f"   = {(tot / t[:, 3]).mean():.0f} per step, {tot.mean():.0f} per tile")
