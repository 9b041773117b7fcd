#!/usr/bin/env python3
"""rehearsal of the multi-rank domain with several gloo ranks on ONE GPU (the collectives are host-staged and the ranks
share the GPU, so only the relative phase costs mean anything): CSTONE_MR_TIMING=1 torchrun ... tools/mr_bench.py"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--particles", type=float, default=2e7, help="global")
p.add_argument("--syncs", type=int, default=8)
a = p.parse_args()
dist.init_process_group("gloo")
rank, P = dist.get_rank(), dist.get_world_size()
import cstone_amd  # noqa: E402
from bench import DistributedPipeline  # noqa: E402

n = int(a.particles)
ctx = cstone_amd.Context(0)
pipe = DistributedPipeline(ctx, n // P, n, 64, 64, "hilbert", max(64, n // (100 * P)), 64, 42 + rank)
pipe.first_sync()
torch.cuda.synchronize()
dist.barrier()
t0 = time.perf_counter()
for _ in range(a.syncs):
    pipe.step()
torch.cuda.synchronize()
dist.barrier()
dt = (time.perf_counter() - t0) / a.syncs
if rank == 0:
    print(f"{P} gloo ranks on one GPU, {n:.1e} particles: {dt*1e3:.2f} ms per sync; rank 0: assigned {pipe.assigned}, "
          f"halos {pipe.halos}, {pipe.stats}", flush=True)
del pipe
dist.destroy_process_group()
