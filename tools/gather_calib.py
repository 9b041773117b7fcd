#!/usr/bin/env python3
"""calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the gather kernels' access pattern (8-byte elements through a
32-bit map; MI355X_MICROARCH.md: only wide coalesced streams are calibrated): cstone_hip_gather_multi of three f64 arrays
of 1e8 elements through (1) the identity -- known bytes, 28 read + 24 written per element --, (2) a random permutation
inside every block of 64 elements (the map of a drifting time step: everything moves a little, inside its leaf), (3) a
random permutation of everything.  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE): the launches
appear in this order, three of each.  tools/condense_profiles.py reads the result (gather_calib_*)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))
import torch  # noqa: E402

import cstone_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ctx = cstone_amd.Context(0)
g = torch.Generator(device="cuda").manual_seed(1)
src = [torch.rand(n, dtype=torch.float64, device="cuda", generator=g) for _ in range(3)]
dst = [torch.empty_like(t) for t in src]
ident = torch.arange(n, device="cuda", dtype=torch.int32)
blocks = (n + 63) // 64
local = (torch.rand(blocks, 64, device="cuda", generator=g).argsort(dim=1).to(torch.int32)
         + 64 * torch.arange(blocks, device="cuda", dtype=torch.int32)[:, None]).reshape(-1)[:n].clamp_(max=n - 1)
full = torch.randperm(n, device="cuda", generator=g).to(torch.int32)
sp = (C.c_void_p * 3)(*[t.data_ptr() for t in src])
dp = (C.c_void_p * 3)(*[t.data_ptr() for t in dst])
for name, m in (("identity", ident), ("inside blocks of 64", local), ("random", full)):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(3):
        torch.cuda.synchronize()
        ctx.profile_enable(True)
        ctx.profile_reset()
        ctx._chk(ctx.lib.cstone_hip_gather_multi(ctx.h, C.c_int(8), C.c_void_p(m.data_ptr()), C.c_size_t(n), sp, dp,
                                                 C.c_int(3)), "gather_multi")
        ctx.sync()
    ms = ctx.profile_get("gather")[0]
    print(f"{name:22s} {ms:.3f} ms  {52.0 * n / ms / 1e6:.0f} GB/s of the 52 algorithmic bytes per element", flush=True)
