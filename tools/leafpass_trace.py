#!/usr/bin/env python3
"""phase budget of the leaf pass for tiles with movers (leafSortBucketsKernel) from a CSTONE_RESORT_TRACE build:
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
t = t[(t[:, :11] > 0).all(axis=1)][:, :11].astype(np.int64)
names = ["setup loads+barrier", "quiet vote", "key loads", "search+atomics", "barrier", "prefix+barrier",
         "scatter+barrier", "scan+stores batch 0", "other batches", "stores drain"]
d = np.diff(t, axis=1) * 10.0 / 1e3  # us (wall_clock64: 100 MHz)
life = (t[:, 10] - t[:, 0]) * 10.0 / 1e3
span = (t[:, 10].max() - t[:, 0].min()) * 10.0 / 1e3
print(f"{kind}: {t.shape[0]} tiles traced, kernel span {span:.1f} us, tile lifetime mean {life.mean():.2f} us "
      f"(p50 {np.median(life):.2f}, p95 {np.percentile(life, 95):.2f}), mean tiles alive {life.sum() / span:.0f}")
print("   median us: " + "  ".join(f"{a} {v:.2f}" for a, v in zip(names, np.median(d, axis=0))))
print("   mean   us: " + "  ".join(f"{a} {v:.2f}" for a, v in zip(names, d.mean(axis=0))))
