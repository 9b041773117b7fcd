#!/usr/bin/env python3
"""micro-benchmark of cstone_hip_find_neighbors on a sorted cloud with its own cornerstone tree
   (BASELINE.json configs[2]: Plummer sphere, bucketSize 64, about 100 neighbours inside 2h)"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))

import torch  # noqa: E402

import cstone_amd  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--n", type=float, default=1e7)
p.add_argument("--dist", default="plummer", choices=["plummer", "uniform"])
p.add_argument("--bucket", type=int, default=64)
p.add_argument("--ngmax", type=int, default=0, help="0: counts only")
p.add_argument("--targets", type=float, default=0, help="number of target particles (0: all)")
p.add_argument("--real-bits", type=int, default=64)
p.add_argument("--ng0", type=float, default=100.0)
p.add_argument("--reps", type=int, default=2)
p.add_argument("--groups", type=float, default=0, help="> 0: targets come from compute_group_splits(64, tol = this)")
a = p.parse_args()
n = int(a.n)
dev = "cuda"
ctx = cstone_amd.Context(0)
rdt = torch.float64 if a.real_bits == 64 else torch.float32
g = torch.Generator(device=dev).manual_seed(3)
if a.dist == "uniform":
    x, y, z = [torch.rand(n, dtype=rdt, device=dev, generator=g) for _ in range(3)]
    h = torch.full((n,), 0.5 * (3.0 * a.ng0 / (4 * math.pi * n)) ** (1 / 3), dtype=rdt, device=dev)
    lim = [0, 1] * 3
else:
    # Plummer sphere of scale radius 1, cut at r = 10: r = (u^(-2/3) - 1)^(-1/2)
    u = torch.rand(n, dtype=torch.float64, device=dev, generator=g).clamp_(1e-12, 1.0)
    r = (u.pow(-2.0 / 3.0) - 1.0).clamp_min_(1e-12).rsqrt().clamp_(max=10.0)
    ct = 2 * torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 1
    ph = 2 * math.pi * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
    st = (1 - ct * ct).sqrt()
    x, y, z = (r * st * ph.cos()).to(rdt), (r * st * ph.sin()).to(rdt), (r * ct).to(rdt)
    rho = 3.0 * n / (4 * math.pi) * (1 + r * r).pow(-2.5)  # number density
    h = (0.5 * (3.0 * a.ng0 / (4 * math.pi * rho)).pow(1 / 3)).clamp_(max=1.0).to(rdt)
    del u, r, ct, ph, st, rho
    lim = [-10.001, 10.001] * 3
box = cstone_amd.make_cbox(lim)
keys = ctx.compute_sfc_keys(cstone_amd.HILBERT, 64, x, y, z, box)
order = torch.empty(n, dtype=torch.int32, device=dev)
ctx.sequence(order)
ctx.sort_pairs(keys, order)
xs, ys, zs, hs = [torch.empty_like(v) for v in (x, y, z, h)]
for src, dst in ((x, xs), (y, ys), (z, zs), (h, hs)):
    ctx.gather(order, src, dst)
del x, y, z, h, order
tree, counts, _ = ctx.compute_octree(keys, a.bucket)
nl = counts.numel()
oc = ctx.build_octree(tree, num_leaves=nl)
cen, siz = ctx.node_centers(cstone_amd.HILBERT, oc["prefixes"], box, a.real_bits)
layout = torch.zeros(nl + 1, dtype=torch.int32, device=dev)
ctx.inclusive_scan(counts, layout[1:])
nt = int(a.targets) or n
first = (n - nt) // 2
groups = None
if a.groups > 0:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    groups = ctx.compute_group_splits(first, first + nt, xs, ys, zs, tree, layout, box, 64, a.groups)
    torch.cuda.synchronize()
    dtg = time.perf_counter() - t0
    t0 = time.perf_counter()
    groups = ctx.compute_group_splits(first, first + nt, xs, ys, zs, tree, layout, box, 64, a.groups)
    torch.cuda.synchronize()
    dtg = time.perf_counter() - t0
    sz = (groups[1:] - groups[:-1]).double()
    print(f"group splits tol {a.groups}: {groups.numel() - 1} groups (fixed: {(nt + 63) // 64}), mean size "
          f"{sz.mean().item():.1f}, {dtg*1e3:.2f} ms", flush=True)
ctx.profile_enable(True)
for rep in range(a.reps + 1):
    if rep == 1:
        ctx.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if groups is None:
        nidx, nc = ctx.find_neighbors(xs, ys, zs, hs, first, first + nt, box, oc, layout, cen, siz, a.ngmax)
    else:
        nidx, nc = ctx.find_neighbors_groups(xs, ys, zs, hs, first, first + nt, groups, box, oc, layout, cen, siz,
                                             a.ngmax)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rep:
        print(f"rep {rep}: {dt*1e3:.2f} ms  {nt/dt/1e6:.2f} M targets/s", flush=True)
ctx.sync()
ms, cnt = ctx.profile_get("neighbors")
ncf = nc.double()
print(f"{a.dist} n={n} targets={nt} leaves={nl} ngmax={a.ngmax}: kernel {ms/cnt:.2f} ms -> {nt/(ms/cnt*1e-3)/1e6:.2f} M targets/s; "
      f"neighbours mean {ncf.mean().item():.1f} max {int(ncf.max().item())}")
