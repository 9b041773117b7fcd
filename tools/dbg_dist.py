#!/usr/bin/env python3
"""diagnostic: DistributedDomain with one rank at bench size; reports where the assigned count changes"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)
import cstone_amd  # noqa: E402
from bench import DistributedPipeline  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
ctx = cstone_amd.Context(0)
pipe = DistributedPipeline(ctx, n, n, 64, 64, "hilbert", max(64, n // 100), 64, 42)
for s in range(6):
    pipe.step()
    print(s, "assigned", pipe.assigned, dict(pipe.dom.stats), "lim", pipe.dom.lim.tolist(), flush=True)
    if pipe.assigned != n:
        x, y, z = pipe.x, pipe.y, pipe.z
        print("min/max", [(float(a.min()), float(a.max())) for a in (x, y, z)])
        break
