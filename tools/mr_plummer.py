#!/usr/bin/env python3
"""the multi-rank sync (RCCL world of one rank) on a Plummer sphere -- BASELINE configs[4]'s kind of cloud at one rank's
share: a deep, very uneven tree.  Checks after every sync what the reference's multi-rank tests check first (keys sorted
and consistent with the coordinates next to them, nobody lost, bucket bound on the rank's own leaves) and prints the time
per steady-state sync with every particle drifting.  usage: mr_plummer.py [particles] [syncs]"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29588")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import cstone_amd  # noqa: E402
from cstone_amd.distributed import NativeDistributedDomain, RcclCollectives  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
syncs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = cstone_amd.Context(0)
g = torch.Generator(device="cuda").manual_seed(4)
u = torch.rand(n, dtype=torch.float64, device="cuda", generator=g).clamp_(1e-12, 1.0)
r = (u.pow(-2.0 / 3.0) - 1.0).clamp_min_(1e-12).rsqrt().clamp_(max=10.0)
ct = 2 * torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 1
ph = 2 * math.pi * torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
st = (1 - ct * ct).sqrt()
x, y, z = r * st * ph.cos(), r * st * ph.sin(), r * ct
rho = 3.0 * n / (4 * math.pi) * (1 + r * r).pow(-2.5)
h = (0.5 * (3.0 * 100.0 / (4 * math.pi * rho)).pow(1 / 3)).clamp_(max=1.0)
del u, r, ct, ph, st, rho
lim = [-10.001, 10.001] * 3
dom = NativeDistributedDomain(ctx, cstone_amd.HILBERT, 64, 64, max(64, n // 100), 64, lim, (0, 0, 0),
                              coll=RcclCollectives(ctx, dist.group.WORLD))
total = 0.0
for s in range(syncs):
    if s:
        for a in (x, y, z):
            a.add_((torch.rand(n, dtype=a.dtype, device="cuda", generator=g) - 0.5) * 0.2 * h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = dom.sync(x, y, z, h)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if s >= 2:
        total += dt
    a, b = res["start"], res["end"]
    keys = res["keys"][a:b]
    v = dom.view()
    ok = b - a == n and bool((keys[1:] >= keys[:-1]).all())
    box = cstone_amd.make_cbox([float(q) for q in res["lim"]], (0, 0, 0))
    again = ctx.compute_sfc_keys(cstone_amd.HILBERT, 64, res["x"][a:b].contiguous(), res["y"][a:b].contiguous(),
                                 res["z"][a:b].contiguous(), box)
    ok = ok and bool(torch.equal(again, keys))
    L = v.num_focus_leaves
    print(f"sync {s}: {dt * 1e3:.2f} ms, focus leaves {L}, re-sorted syncs so far {v.resorts}, keys sorted and consistent: {ok}",
          flush=True)
    assert ok
    x, y, z, h = [res[k][a:b].clone() for k in "xyzh"]
print(f"Plummer sphere, {n:.1e} particles on one rank: {total / max(1, syncs - 2) * 1e3:.2f} ms per steady-state sync")
del dom
dist.destroy_process_group()
