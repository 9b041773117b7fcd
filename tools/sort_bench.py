#!/usr/bin/env python3
"""micro-benchmark of cstone_hip_sort_pairs (used under rocprofv3 for the per-kernel counters in profiles/)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))

import torch  # noqa: E402

import cstone_amd  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--n", type=float, default=1e8)
p.add_argument("--key-bits", type=int, default=64)
p.add_argument("--reps", type=int, default=3)
p.add_argument("--sorted", action="store_true", help="input already sorted (steady-state sync)")
p.add_argument("--alt-offset", type=int, default=0, help="byte offset of the alternate buffers inside their allocations "
                                                         "(do the read and the write stream collide in the DRAM banks?)")
a = p.parse_args()
n = int(a.n)
ctx = cstone_amd.Context(0)
g = torch.Generator(device="cuda").manual_seed(1)
if a.key_bits == 64:
    src = torch.randint(0, 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=g)
else:
    src = torch.randint(0, 2**30, (n,), dtype=torch.int32, device="cuda", generator=g)
if a.sorted:
    src = torch.sort(src).values
keys = torch.empty_like(src)
vals = torch.empty(n, dtype=torch.int32, device="cuda")
if a.alt_offset:
    kraw = torch.empty(keys.numel() * keys.element_size() + a.alt_offset, dtype=torch.uint8, device="cuda")
    vraw = torch.empty(vals.numel() * 4 + a.alt_offset, dtype=torch.uint8, device="cuda")
    ka = kraw[a.alt_offset:].view(keys.dtype)
    va = vraw[a.alt_offset:].view(torch.int32)
else:
    ka, va = torch.empty_like(keys), torch.empty_like(vals)
tmp = torch.empty(ctx.sort_temp_bytes(a.key_bits, n), dtype=torch.uint8, device="cuda")
ctx.profile_enable(True)
for rep in range(a.reps + 1):
    keys.copy_(src)
    ctx.sequence(vals)
    if rep == 1:
        ctx.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.sort_pairs(keys, vals, ka, va, tmp)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rep:
        print(f"rep {rep}: {dt*1e3:.3f} ms  {n/dt/1e9:.3f} Gpairs/s")
ms, cnt = ctx.profile_get("sort_pass")
hms, hc = ctx.profile_get("sort_hist")
kb = a.key_bits // 8
print(f"pass: {ms/cnt:.4f} ms avg over {cnt} launches -> {2*(kb+4)*n/(ms/cnt*1e-3)/1e9:.1f} GB/s algorithmic; "
      f"hist: {hms/hc:.4f} ms")
ctx.sync()
assert bool((keys[1:] >= keys[:-1]).all())
