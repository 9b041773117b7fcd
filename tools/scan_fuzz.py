#!/usr/bin/env python3
"""fuzz of the device scans (cstone_hip_exclusive_scan_u32 / _inclusive_scan_u32) against torch.cumsum: many sizes in a row on
ONE context, in place and out of place -- what a long-lived client does to the single-launch scan's persistent state"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))
import torch  # noqa: E402

import cstone_amd  # noqa: E402

ctx = cstone_amd.Context(0)
g = torch.Generator(device="cuda").manual_seed(5)
bad = 0
sizes = [1, 2, 63, 64, 65, 2047, 2048, 2049, 4096, 4097, 20000, 20001, 65536, 100000, 300000, 524288, 524289, 600000]
import random
random.seed(3)
for it in range(3000):
    n = random.choice(sizes) if it % 3 else random.randint(1, 530000)
    a = torch.randint(0, 50, (n,), dtype=torch.int32, device="cuda", generator=g)
    want_ex = torch.cumsum(a.long(), 0) - a.long()
    mode = it % 4
    if mode == 0:
        out = torch.empty_like(a)
        rc = ctx.lib.cstone_hip_exclusive_scan_u32(ctx.h, C.c_void_p(a.data_ptr()), C.c_void_p(out.data_ptr()), C.c_size_t(n), C.c_uint32(7))
        ok = bool((out.long() == want_ex + 7).all())
    elif mode == 1:
        out = a.clone()
        rc = ctx.lib.cstone_hip_exclusive_scan_u32(ctx.h, C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr()), C.c_size_t(n), C.c_uint32(0))
        ok = bool((out.long() == want_ex).all())
    elif mode == 2:
        out = torch.empty_like(a)
        rc = ctx.lib.cstone_hip_inclusive_scan_u32(ctx.h, C.c_void_p(a.data_ptr()), C.c_void_p(out.data_ptr()), C.c_size_t(n))
        ok = bool((out.long() == want_ex + a.long()).all())
    else:
        out = a.clone()
        rc = ctx.lib.cstone_hip_inclusive_scan_u32(ctx.h, C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr()), C.c_size_t(n))
        ok = bool((out.long() == want_ex + a.long()).all())
    if rc != 0 or not ok:
        bad += 1
        if bad < 10:
            d = (out.long() != (want_ex + (7 if mode == 0 else 0) + (a.long() if mode >= 2 else 0))).nonzero()
            print("MISMATCH it", it, "n", n, "mode", mode, "rc", rc, "first bad", int(d[0]) if d.numel() else None, "count", d.numel(), flush=True)
ctx.sync()
print("scan fuzz:", "OK" if bad == 0 else f"{bad} failures")
sys.exit(0 if bad == 0 else 1)
