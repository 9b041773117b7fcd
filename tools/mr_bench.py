#!/usr/bin/env python3
"""rehearsal of the multi-rank domain on ONE GPU: several gloo ranks (collectives host-staged, ranks share the GPU: only the
relative phase costs mean anything; CSTONE_MR_TIMING=1 torchrun ... tools/mr_bench.py) or, with --rccl and no launcher,
a world of ONE rank whose collectives are served by RCCL from C++ inside the library (the workload of
profiles/r02_mr_sync_kernels.json: the per-GPU share of the 8-GPU point is --particles 1.25e7)"""
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
p.add_argument("--dist", default="uniform", choices=["uniform", "plummer", "clustered"])
p.add_argument("--rccl", action="store_true", help="backend nccl: RCCL collectives from inside libcstone_hip")
a = p.parse_args()
if a.rccl:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo")
rank, P = dist.get_rank(), dist.get_world_size()
import cstone_amd  # noqa: E402
from bench import DistributedPipeline  # noqa: E402

n = int(a.particles)
ctx = cstone_amd.Context(0)
pipe = DistributedPipeline(ctx, n // P, n, 64, 64, "hilbert", max(64, n // (100 * P)), 64, seed=42, dist=a.dist, rank=rank,
                           world=P, backend="nccl" if a.rccl else "gloo")
pipe.first_sync()
for _ in range(2):
    pipe.drift()
    pipe.step()
dt = 0.0
for _ in range(a.syncs):
    pipe.drift()  # every particle by up to 0.1 h per coordinate, outside the timed intervals (as bench.py)
    torch.cuda.synchronize()
    if P > 1:  # (a barrier of one rank is a fill kernel and a stream synchronisation of its own in the API trace)
        dist.barrier()
    t0 = time.perf_counter()
    pipe.step()
    torch.cuda.synchronize()
    if P > 1:
        dist.barrier()
    dt += time.perf_counter() - t0
dt /= a.syncs
if rank == 0:
    print(f"{P} {'RCCL' if a.rccl else 'gloo'} rank(s) on one GPU, {n:.3g} particles: {dt*1e3:.2f} ms per sync; rank 0: assigned {pipe.assigned}, "
          f"halos {pipe.halos}, {pipe.stats}, syncs re-sorted: {pipe.dom.view().resorts}", flush=True)
del pipe
dist.destroy_process_group()
