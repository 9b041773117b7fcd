#!/usr/bin/env python3
"""Worker of the multi-rank tests (launched with torch.distributed.run, backend gloo).
--backend cpu : oracle-based CPU backend (no GPU needed)      --backend hip : libcstone_hip on cuda:0 (ranks share it)
Every rank owns a random 1/P of a global cloud, runs DistributedDomain.sync `steps` times while the particles move and
checks, like the reference's own multi-rank test (test/integration_mpi/domain_nranks.cpp:80-150):
  * the assigned particle counts add up to N and every rank's keys are sorted and inside its SFC range,
  * the sum over ranks of the neighbour counts found with LOCAL + HALO particles equals the neighbour count sum of
    the undistributed cloud (i.e. every neighbour within 2h of an assigned particle is present)."""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def tag64_of(r, st, en):
    return r["x"][st:en] * 3.0 + r["y"][st:en] * 5.0 + r["z"][st:en] * 7.0 + r["h"][st:en]


def neighbor_sum(o, x, y, z, h, first, last, lim, bc):
    """brute-force-free reference count through the oracle's tree search on the given particle set"""
    from oracle.oracle import HILBERT, Box

    box = Box(lim, bc)
    keys = o.compute_sfc_keys(HILBERT, 64, x, y, z, box)
    ks, order = o.sort_pairs(keys, np.arange(x.size))
    inv = np.empty_like(order)
    inv[order] = np.arange(x.size, dtype=order.dtype)
    xs, ys, zs, hs = x[order], y[order], z[order], h[order]
    tree, counts = o.compute_octree(ks, 32)
    oc = o.build_octree(tree)
    layout = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    cen, siz = o.node_centers(HILBERT, oc["prefixes"], box, x.dtype.itemsize * 8)
    _, nc = o.find_neighbors(xs, ys, zs, hs, 0, x.size, box, oc, layout, cen, siz, 1)
    sel = inv[first:last]
    return int(nc[sel].astype(np.int64).sum())


def late_fields(x, y, z, h):
    """fields of 8, 12, 1 and 32 bytes per particle that are NOT passed to sync: reapplySync must route them afterwards"""
    return [x * 11.0 - z, torch.stack([x, y, z], dim=1).to(torch.float32).contiguous(),
            (x * 200.0).to(torch.uint8), torch.stack([x, y, z, h], dim=1).contiguous()]


def check_reapply(dom, late, r):
    """Domain::reapplySync (R/domain/domain.hpp:334-378): the assigned range holds the values of its particles"""
    st, en = r["start"], r["end"]
    want = late_fields(*[r[k][st:en] for k in "xyzh"])
    ok = True
    for f, w in zip(late, want):
        out = dom.reapply_sync(f)
        ok &= out.shape[0] == r["x"].numel() and bool(torch.equal(out[st:en], w))
    if late[0].shape[0] == 0:
        return ok
    try:
        dom.reapply_sync(late[0][:-1])  # checkSizesEqual: an array that does not match the last sync's input
        ok = False
    except Exception:
        pass
    return ok


def make_native(backend, bucket, bucket_focus, lim, bc, curve=1, key_bits=64, real_bits=64, owner_side=0):
    from cstone_amd.distributed import NativeDistributedDomain

    return NativeDistributedDomain(backend.ctx, curve, key_bits, real_bits, bucket, bucket_focus, lim, bc,
                                   halo_mode=NativeDistributedDomain.HALOS_OWNER_SIDE if owner_side else None)


def golden(a, backend, dev, rank, P):
    """the reference's Domain on P MPI ranks (fixture) against DistributedDomain on P torch.distributed ranks: box,
    SFC ranges, global tree + counts and the assigned particles (keys, x, h) of every rank after every sync, bit for bit"""
    from py_domain import Comm, DistributedDomain
    from oracle import oracle as orc

    g = np.load(a.golden)
    assert int(g["P"]) == P, "launch with the fixture's number of ranks"
    mine = np.nonzero(g["owner"] == rank)[0]
    x, y, z, h = [torch.from_numpy(g[k][mine].copy()).to(dev) for k in "xyzh"]
    kb = int(g["key_bits"]) if "key_bits" in g else 64  # the instantiation Domain<KeyType, T> the fixture was made with
    rb = int(g["real_bits"]) if "real_bits" in g else 64
    kdt = np.uint64 if kb == 64 else np.uint32
    assert x.element_size() * 8 == rb
    if a.impl == "native":
        dom = make_native(backend, int(g["bucket"]), int(g["bucket_focus"]), g["lim"].tolist(),
                          tuple(int(v) for v in g["bc"]), key_bits=kb, real_bits=rb, owner_side=a.owner_side)
    else:
        dom = DistributedDomain(backend, Comm(), orc.HILBERT, kb, rb, bucket=int(g["bucket"]),
                                bucket_focus=int(g["bucket_focus"]), box_lim=g["lim"].tolist(),
                                box_bc=tuple(int(v) for v in g["bc"]))
    bad, halo_stats = [], []
    # the motion of oracle/ref_domain_mpi.cpp, in the fixture's real type (separately rounded operations)
    c, top = 0.01, (1.0 - 2.0**-30 if rb == 64 else float(np.float32(1.0 - 2.0**-20)))
    for s in range(int(g["syncs"])):
        r = dom.sync(x, y, z, h)
        st, en = r["start"], r["end"]
        keys = r["keys"].cpu().numpy().view(kdt)[st:en]
        if a.impl == "native":
            v = dom.view()
            rng_ = [v.range_start, v.range_end]
            gl = dom.fetch(v.global_leaves, v.num_global_leaves + 1, kdt)
            gc = dom.fetch(v.global_counts, v.num_global_leaves, np.uint32)
        else:
            L = dom.g_leaves
            rng_ = [dom.assignment[rank], dom.assignment[rank + 1]]
            gl = backend.keys_to_numpy(dom.gtree[:L + 1], kb)
            gc = backend.to_numpy(dom.gcounts[:L]).view(np.uint32)
        ties_any = "dups" in os.path.basename(a.golden)  # equal keys: any order among them is the reference's order

        def same(field, want):
            got = r[field][st:en].cpu().numpy()
            if not ties_any:
                return np.array_equal(got, want)
            wk = g[f"s{s}_r{rank}_keys"]
            return keys.size == wk.size and np.array_equal(got[np.lexsort((got, keys))], want[np.lexsort((want, wk))])

        checks = {
            "lim": np.array_equal(r["lim"], g[f"s{s}_r{rank}_lim"]),
            "range": rng_ == [int(v) for v in g[f"s{s}_r{rank}_range"]],
            "leaves": np.array_equal(gl, g[f"s{s}_leaves"]),
            "counts": np.array_equal(gc, g[f"s{s}_counts"]),
            "keys": np.array_equal(keys, g[f"s{s}_r{rank}_keys"]),
            "x": same("x", g[f"s{s}_r{rank}_x"]),
            "h": same("h", g[f"s{s}_r{rank}_h"]),
        }
        hx, hy, hz = [np.concatenate([r[k][:st].cpu().numpy(), r[k][en:].cpu().numpy()]) for k in "xyz"]
        got = set(zip(hx.tolist(), hy.tolist(), hz.tolist()))
        ref = set(zip(*[v.tolist() for v in g[f"s{s}_r{rank}_halos"]]))
        extra, missing = len(got - ref), len(ref - got)
        halo_stats.append((len(got), len(ref), extra, missing))
        if a.impl == "native" and not a.owner_side:
            # the library's locally essential tree (csrc/let.hpp) IS the reference's focus tree: leaves, counts, the
            # rank's cells, layout(), nParticlesWithHalos() and the halo particles in buffer order, bit for bit
            L = v.num_focus_leaves
            hk = np.concatenate([r["keys"].cpu().numpy().view(kdt)[:st], r["keys"].cpu().numpy().view(kdt)[en:]])
            hh = np.concatenate([r["h"][:st].cpu().numpy(), r["h"][en:].cpu().numpy()])
            checks.update({
                "start / end / size": [st, en, int(r["x"].numel())] == [int(t) for t in g[f"s{s}_r{rank}_info"][:3]],
                "focus leaves": np.array_equal(dom.fetch(v.focus_leaves, L + 1, kdt), g[f"s{s}_r{rank}_focus_leaves"]),
                "focus counts": np.array_equal(dom.fetch(v.focus_leaf_counts, L, np.uint32),
                                               g[f"s{s}_r{rank}_focus_counts"]),
                "cells": [v.start_cell, v.end_cell] == [int(t) for t in g[f"s{s}_r{rank}_cells"]],
                "layout": np.array_equal(dom.fetch(v.layout, L + 1, np.uint32), g[f"s{s}_r{rank}_layout"]),
                "halo keys": np.array_equal(hk, g[f"s{s}_r{rank}_halo_keys"]),
            })
            want = g[f"s{s}_r{rank}_halos"]
            if ties_any:  # equal keys: the order among them is the owner's business (MPI_ANY_SOURCE in the reference)
                o1, o2 = np.lexsort((hz, hy, hx, hk)), np.lexsort((want[2], want[1], want[0], g[f"s{s}_r{rank}_halo_keys"]))
                checks["halo particles"] = all(np.array_equal(p[o1], q[o2]) for p, q in
                                               zip((hx, hy, hz), (want[0], want[1], want[2])))
            else:
                checks["halo particles"] = all(np.array_equal(p, q) for p, q in
                                               zip((hx, hy, hz, hh), (want[0], want[1], want[2], g[f"s{s}_r{rank}_halo_h"])))
        elif a.impl == "native":
            # owner-side discovery (opt-in): complete, no particle that the reference does not have as well in the
            # fixtures with distinct keys; a few cells may resolve differently
            checks["halo set (owner-side discovery)"] = len(got) == hx.size and extra + missing <= 0.02 * max(1, len(ref))
        bad += [f"sync {s} rank {rank}: {k}" for k, ok in checks.items() if not ok]
        xo, yo, zo = [r[k][st:en].clone() for k in "xyz"]
        h = r["h"][st:en].clone()
        x = torch.clamp(xo + c * (yo - 0.5), 0.0, top)
        y = torch.clamp(yo + c * (zo - 0.5), 0.0, top)
        z = torch.clamp(zo + c * (xo - 0.5), 0.0, top)
    allbad, allhalos = [None] * P, [None] * P
    dist.all_gather_object(allbad, bad)
    dist.all_gather_object(allhalos, halo_stats)
    flat = [b for part in allbad for b in part]
    if rank == 0:
        print("DIST_RESULT " + json.dumps(dict(ok=not flat, ranks=P, mismatches=flat[:20], report=[],
                                               halos_found_vs_reference=allhalos)))
    dist.destroy_process_group()
    return 0 if not flat else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="cpu")
    ap.add_argument("--particles", type=int, default=20000)
    ap.add_argument("--syncs", type=int, default=3)
    ap.add_argument("--pbc", type=int, default=0)
    ap.add_argument("--golden", default="", help="fixture of tests/golden/make_golden_domain_mpi.py to reproduce")
    ap.add_argument("--key-bits", type=int, default=64)
    ap.add_argument("--real-bits", type=int, default=64)
    ap.add_argument("--curve", default="hilbert", choices=["hilbert", "morton"])
    ap.add_argument("--lopsided", type=int, default=0, help="1: the last rank starts without particles")
    ap.add_argument("--impl", default="python", choices=["python", "native"],
                    help="python: tests/py_domain.DistributedDomain; native: cstone_hip_domain_mr_* (hip only)")
    ap.add_argument("--contract", type=int, default=0,
                    help="native only: this many more syncs in which the cloud contracts by 3.5 % (the trees deepen)")
    ap.add_argument("--owner-side", type=int, default=0,
                    help="native only: 1 = owner-side halo discovery instead of the locally essential tree")
    ap.add_argument("--fail-at", default="", help="native only: after one good sync, rank 1 is made to fail at this point "
                                                  "of the next sync (CSTONE_MR_FAIL_AT); every rank must get an error")
    ap.add_argument("--grav", type=int, default=0,
                    help="native only: this many Domain::syncGrav calls (cstone_hip_domain_mr_sync_grav): masses follow their "
                         "particles, the root's expansion centre is the global centre of mass, neighbour counts complete")
    ap.add_argument("--spec-box", type=int, default=0,
                    help="native only: this many syncs through two domains, one encoding with the previous box while it "
                         "measures the extents (the default), one measuring first (CSTONE_NO_SPECULATIVE_BOX); particles "
                         "jitter inside fixed extents (the box holds) except for two steps that push them outwards")
    a = ap.parse_args()
    dist.init_process_group("gloo")
    rank, P = dist.get_rank(), dist.get_world_size()

    from py_domain import Comm, DistributedDomain, HipBackend
    from oracle import oracle as orc

    o = orc.Oracle()
    if a.backend == "hip":
        import cstone_amd

        torch.cuda.set_device(0)
        backend = HipBackend(cstone_amd.Context(0))
        dev = "cuda"
    else:
        from cpu_backend import CpuBackend

        backend = CpuBackend()
        dev = "cpu"

    if a.golden:
        sys.exit(golden(a, backend, dev, rank, P))

    N = a.particles
    rng = np.random.default_rng(2024)
    lim = [0.0, 1.0, 0.0, 1.0, 0.0, 1.0]
    bc = (1, 1, 1) if a.pbc else (0, 0, 0)
    centers = rng.uniform(0.2, 0.8, (4, 3))
    pos = np.where(rng.uniform(size=(N, 1)) < 0.5, rng.uniform(0, 1, (N, 3)),
                   centers[rng.integers(0, 4, N)] + rng.normal(0, 0.05, (N, 3)))
    pos = np.clip(pos, 0.0, 1.0 - 1e-9)
    hglob = 0.035 * rng.uniform(0.6, 1.2, N)
    owner = rng.integers(0, P, N)  # random initial ownership: worst-case first exchange
    if a.lopsided and P > 1:
        owner = owner % (P - 1)  # the last rank brings nothing to the first sync
    vel = rng.normal(0, 0.004, (N, 3))

    mine = np.nonzero(owner == rank)[0]
    ids = mine.copy()
    rdt = torch.float64 if a.real_bits == 64 else torch.float32
    top = 1.0 - 1e-9 if a.real_bits == 64 else 1.0 - 2.0**-20
    x, y, z = [torch.from_numpy(pos[mine, d].copy()).to(dev).to(rdt).clamp_(0.0, top) for d in range(3)]
    h = torch.from_numpy(hglob[mine].copy()).to(dev).to(rdt)

    curve = orc.HILBERT if a.curve == "hilbert" else orc.MORTON
    kdt = np.uint64 if a.key_bits == 64 else np.uint32
    if a.impl == "native":
        dom = make_native(backend, max(64, N // (100 * P)), 16, lim, bc, curve, a.key_bits, a.real_bits, a.owner_side)
    else:
        dom = DistributedDomain(backend, Comm(), curve, a.key_bits, a.real_bits, bucket=max(64, N // (100 * P)),
                                bucket_focus=16, box_lim=lim, box_bc=bc)
    ok = True
    report = []
    if a.spec_box:
        # the speculative box of the multi-rank sync (domain_mr.hip): same results as measuring first, whether the box
        # holds (most steps here) or not (steps 3 and 4), on every rank
        os.environ["CSTONE_NO_SPECULATIVE_BOX"] = "1"
        dom_b = make_native(backend, max(64, N // (100 * P)), 16, lim, bc, curve, a.key_bits, a.real_bits, a.owner_side)
        del os.environ["CSTONE_NO_SPECULATIVE_BOX"]
        dom_b.set_sort_mode(dom_b.SORT_ALL_DIGITS)  # ... and never re-sorts (cstone_hip_domain_mr_set_sort_mode)
        g = torch.Generator(device=dev).manual_seed(77 + rank)
        xa, ya, za, ha = x, y, z, h
        for s_ in range(a.spec_box):
            ra = dom.sync(xa, ya, za, ha)
            rb = dom_b.sync(xa.clone(), ya.clone(), za.clone(), ha.clone())
            same = ra["start"] == rb["start"] and ra["end"] == rb["end"] and list(ra["lim"]) == list(rb["lim"])
            for k in ("keys", "x", "y", "z", "h"):
                same = same and bool(torch.equal(ra[k], rb[k]))
            va, vb = dom.view(), dom_b.view()
            same = same and (va.num_focus_leaves, va.start_cell, va.end_cell, va.num_global_leaves) == (
                vb.num_focus_leaves, vb.start_cell, vb.end_cell, vb.num_global_leaves)
            ok &= same
            report.append(dict(sync=s_, same=same, lim=[float(v) for v in ra["lim"]], resorts=int(va.resorts)))
            st, en = ra["start"], ra["end"]
            xa, ya, za, ha = [ra[k][st:en].clone() for k in "xyzh"]
            if s_ in (2, 3):   # everybody drifts outwards a little: all extents move, on every rank
                for c in (xa, ya, za):
                    c.sub_(0.5).mul_(1.002).add_(0.5)
            else:              # jitter inside the global extents of the cloud: the box stays what it is
                for c, d in zip((xa, ya, za), range(3)):
                    lo, hi = float(ra["lim"][2 * d]), float(ra["lim"][2 * d + 1])
                    keep = (c <= lo) | (c >= hi)  # (the particles that define the extents stay where they are)
                    moved = c + (torch.rand(c.numel(), dtype=c.dtype, device=c.device, generator=g) - 0.5) * 2e-4
                    c.copy_(torch.where(keep, c, moved.clamp_(lo, hi)))
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
        # the speculating domain re-sorted where the box held (at least two of the quiet steps), the other one never
        ok = ok and int(dom.view().resorts) >= 2 and int(dom_b.view().resorts) == 0
        if rank == 0:
            print("DIST_RESULT " + json.dumps(dict(ok=ok, ranks=P, report=report)))
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)
    if a.grav:
        # Domain::syncGrav on the RCCL route (the LET state machine itself is compared with the reference in
        # oracle/let_check.cpp `grav`; here: the entry point end to end on several ranks)
        mass = torch.from_numpy((0.5 + 10.0 * hglob[mine]) / N).to(dev).to(rdt if a.real_bits == 64 else torch.float32)
        xa, ya, za, ha, ma = x, y, z, h, mass
        for s_ in range(a.grav):
            ident = xa * 3.0 + ya * 5.0 + za * 7.0 + ha
            r = dom.sync_grav(xa, ya, za, ha, ma, props=[ident])
            st, en = r["start"], r["end"]
            same = bool(torch.equal(r["props"][0][st:en], (r["x"] * 3.0 + r["y"] * 5.0 + r["z"] * 7.0 + r["h"])[st:en]))
            k = r["keys"]
            same = same and (bool((k[1:] >= k[:-1]).all()) if k.numel() > 1 else True)
            # masses stayed attached: m is a function of h here
            want_m = ((0.5 + 10.0 * r["h"][st:en].double()) / N).to(r["m"].dtype)
            same = same and bool(torch.allclose(r["m"][st:en], want_m, rtol=1e-6, atol=0))
            oc = dom.octree()
            ec = oc["expansion_centers"]
            same = same and ec is not None
            # the root's expansion centre is the centre of mass of the WHOLE cloud (the global centre exchange)
            loc = torch.stack([(r["m"][st:en].double() * r[c][st:en].double()).sum() for c in "xyz"] +
                              [r["m"][st:en].double().sum()]).cpu()
            dist.all_reduce(loc)
            com = (loc[:3] / loc[3]).numpy()
            root = ec[0, :3].double().cpu().numpy()
            close = bool(np.allclose(root, com, rtol=1e-5 if a.real_bits == 32 else 1e-10, atol=1e-7 if a.real_bits == 32 else 1e-12))
            same = same and close and float(ec[:, 3].min()) >= 0.0
            tot = torch.tensor([en - st], dtype=torch.int64)
            dist.all_reduce(tot)
            same = same and int(tot.item()) == N
            ok &= same
            report.append(dict(grav_sync=s_, ok=same, root=[float(v) for v in root], com=[float(v) for v in com],
                               focus_leaves=int(dom.view().num_focus_leaves)))
            xa, ya, za, ha, ma = [r[c][st:en].clone() for c in ("x", "y", "z", "h", "m")]
            for c, vcol in zip((xa, ya, za), range(3)):
                c.add_(0.004 * torch.sin(7.0 * (xa + vcol))).clamp_(0.0, top)
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
        if rank == 0:
            print("DIST_RESULT " + json.dumps(dict(ok=ok, ranks=P, report=report)))
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)
    if a.fail_at:
        # collective-safe failure: one good sync, then rank 1 fails inside the next one; nobody may hang, everybody must
        # see an error, and the domain must work again afterwards
        import cstone_amd

        r = dom.sync(x, y, z, h)
        st, en = r["start"], r["end"]
        nxt = [r[k][st:en].clone() for k in "xyzh"]
        for c in nxt[:3]:  # the cloud contracts: the failing sync rebalances its trees before it fails
            c.sub_(0.5).mul_(0.965).add_(0.5)
        os.environ["CSTONE_MR_FAIL_AT"] = f"1:{a.fail_at}"
        raised, msg = 0, ""
        try:
            dom.sync(*[c.clone() for c in nxt])
        except cstone_amd.CstoneError as e:
            raised, msg = 1, str(e)
        del os.environ["CSTONE_MR_FAIL_AT"]
        flag = torch.tensor([raised])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        named = ("injected" in msg) if rank == 1 else ("rank 1 reported a failure" in msg)
        good = torch.tensor([1 if named else 0])
        dist.all_reduce(good, op=dist.ReduceOp.MIN)
        # the retry: same input sizes and box as the failed sync.  What the failed sync left of its tree must not be what
        # the re-sort starts from: the retry sorts from scratch, the sync after it re-sorts again
        before = int(dom.view().resorts)
        r2 = dom.sync(*[c.clone() for c in nxt])
        retried_from_scratch = int(dom.view().resorts) == before
        k2 = r2["keys"]
        sorted2 = bool((k2[1:] >= k2[:-1]).all()) if k2.numel() > 1 else True
        tot = torch.tensor([r2["end"] - r2["start"]], dtype=torch.int64)
        dist.all_reduce(tot)
        r3 = dom.sync(*[r2[k][r2["start"]:r2["end"]].clone() for k in "xyzh"])
        k3 = r3["keys"]
        sorted3 = bool((k3[1:] >= k3[:-1]).all()) if k3.numel() > 1 else True
        resorted_again = int(dom.view().resorts) == before + 1
        tot3 = torch.tensor([r3["end"] - r3["start"]], dtype=torch.int64)
        dist.all_reduce(tot3)
        mine_ok = torch.tensor([1 if (retried_from_scratch and sorted2 and sorted3 and resorted_again) else 0])
        dist.all_reduce(mine_ok, op=dist.ReduceOp.MIN)
        ok = (bool(flag.item()) and bool(good.item()) and int(tot.item()) == N and int(tot3.item()) == N and
              bool(mine_ok.item()))
        msg = f"{msg} | retry from scratch {retried_from_scratch}, re-sorted afterwards {resorted_again}"
        if rank == 0:
            print("DIST_RESULT " + json.dumps(dict(ok=ok, ranks=P, report=[dict(message=msg)])))
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)
    tag64 = x * 3.0 + y * 5.0 + z * 7.0 + h          # conserved fields that must stay attached to their particles
    tag32 = (x + 2.0 * y).to(torch.float32)
    for s in range(a.syncs):
        if a.impl == "native":
            late = late_fields(x, y, z, h)
            vec3 = torch.stack([x, y, z], dim=1).to(torch.float32).contiguous()  # a Vec3<float> property (12 bytes)
            r = dom.sync(x, y, z, h, props=[tag64, tag32, vec3])
        else:
            r = dom.sync(x, y, z, h)
        st, en = r["start"], r["end"]
        if a.impl == "native":
            ok &= check_reapply(dom, late, r)
            p64, p32, p3 = r["props"]
            ok &= bool(torch.equal(p3[st:en], torch.stack([r[k][st:en] for k in "xyz"], dim=1).to(torch.float32)))
            ok &= bool(torch.equal(p64[st:en], tag64_of(r, st, en)))
            ok &= bool(torch.equal(p32[st:en], (r["x"][st:en] + 2.0 * r["y"][st:en]).to(torch.float32)))
        keys = r["keys"].cpu().numpy().view(kdt)
        # invariants: counts add up, keys sorted, assigned keys inside my range
        tot = torch.tensor([en - st], dtype=torch.int64)
        dist.all_reduce(tot)
        ok &= int(tot.item()) == N
        ok &= bool(np.all(keys[1:] >= keys[:-1]))
        if a.impl == "native":
            v = dom.view()
            lo_key, hi_key, stats = v.range_start, v.range_end, dict(moved=v.particles_sent, halos=v.halos_received,
                                                                      served=v.halos_sent, halo_boxes=v.halo_boxes_exported)
            # exchangeHalos: a field that equals 2x + 1 on the assigned range gets the owners' values in the halo ranges
            for dt in (torch.float64, torch.float32):
                f = torch.full_like(r["x"], -7.0, dtype=dt)  # 8- and 4-byte fields whatever the coordinate type
                f[st:en] = (2.0 * r["x"][st:en] + 1.0).to(dt)
                dom.exchange_halos(f)
                ok &= bool(torch.equal(f, (2.0 * r["x"] + 1.0).to(dt)))
            # a Vec3<double> field (24-byte elements)
            ref3 = torch.stack([r["x"], r["y"], r["z"]], dim=1).contiguous()
            f3 = torch.full_like(ref3, -7.0)
            f3[st:en] = ref3[st:en]
            dom.exchange_halos(f3)
            ok &= bool(torch.equal(f3, ref3))
        else:
            lo_key, hi_key, stats = dom.assignment[rank], dom.assignment[rank + 1], dict(dom.stats)
        ok &= bool(np.all(keys[st:en] >= kdt(lo_key))) and (en == st or int(keys[en - 1]) < hi_key)
        # neighbour completeness
        lx, ly, lz, lh = [r[k].cpu().numpy() for k in "xyzh"]
        local_sum = neighbor_sum(o, lx, ly, lz, lh, st, en, r["lim"], bc)
        if a.impl == "native":
            # Domain::octreeProperties(): the domain's own tree over local + halo particles feeds the neighbor search
            import cstone_amd

            oc = dom.octree()
            lay = oc["layout"]
            ok &= int(lay[0]) == 0 and int(lay[-1]) == r["x"].numel() and oc["num_leaves"] == lay.numel() - 1
            _, nc = backend.ctx.find_neighbors(r["x"], r["y"], r["z"], r["h"], st, en, cstone_amd.make_cbox(r["lim"], bc),
                                               oc, lay, oc["centers"], oc["sizes"], 0)
            ok &= int(nc.long().sum().item()) == local_sum
            # bucketFocus 16: converged on the first sync -- inside the rank's own cells; the locally essential tree is
            # coarser elsewhere (owner-side mode: the tree over local + halo particles resolves everything)
            own = oc["leaf_counts"] if a.owner_side else oc["leaf_counts"][v.start_cell:v.end_cell]
            ok &= int(own.max().item()) <= 16 or s > 0
        tsum = torch.tensor([local_sum], dtype=torch.int64)
        dist.all_reduce(tsum)
        # the undistributed cloud at this step: gather the assigned particles of every rank
        parts = [None] * P
        dist.all_gather_object(parts, np.stack([lx[st:en], ly[st:en], lz[st:en], lh[st:en]]))
        if rank == 0:
            allp = np.concatenate(parts, axis=1)
            ref = neighbor_sum(o, allp[0].copy(), allp[1].copy(), allp[2].copy(), allp[3].copy(), 0, allp.shape[1],
                               r["lim"], bc)
            ok &= ref == int(tsum.item())
            report.append(dict(step=s, neighbors=ref, found=int(tsum.item()), stats=stats))
        # move: assigned particles only (halos are discarded by the client before the next sync)
        m = en - st
        drift = torch.from_numpy(rng.normal(0, 0.004, (m, 3))).to(dev).to(rdt)
        x, y, z, h = [r[k][st:en].clone() for k in "xyzh"]
        x, y, z = x + drift[:, 0], y + drift[:, 1], z + drift[:, 2]
        if a.pbc:
            x, y, z = [torch.remainder(v, 1.0).clamp_(0.0, top) for v in (x, y, z)]
        else:
            x, y, z = x.clamp(0.0, top), y.clamp(0.0, top), z.clamp(0.0, top)
        tag64 = x * 3.0 + y * 5.0 + z * 7.0 + h
        tag32 = (x + 2.0 * y).to(torch.float32)
    if a.impl == "native" and a.contract:
        # the cloud contracts towards the box centre, a few per cent per sync, over many RE-SORTED syncs: the rank's tree
        # deepens by a level every few syncs (across the 5 -> 6 boundary of the node-key sort's digit passes as well);
        # neighbour completeness after every sync shows that the linked octree behind the halo search stayed right
        for extra in range(a.contract):
            x, y, z = [(0.5 + (v_ - 0.5) * 0.965).clamp_(0.0, top) for v_ in (x, y, z)]
            h = h * 0.965
            r = dom.sync(x, y, z, h)
            st, en = r["start"], r["end"]
            tot = torch.tensor([en - st], dtype=torch.int64)
            dist.all_reduce(tot)
            ok &= int(tot.item()) == N
            lx, ly, lz, lh = [r[k].cpu().numpy() for k in "xyzh"]
            tsum = torch.tensor([neighbor_sum(o, lx, ly, lz, lh, st, en, r["lim"], bc)], dtype=torch.int64)
            dist.all_reduce(tsum)
            parts = [None] * P
            dist.all_gather_object(parts, np.stack([lx[st:en], ly[st:en], lz[st:en], lh[st:en]]))
            if rank == 0:
                allp = np.concatenate(parts, axis=1)
                ref = neighbor_sum(o, allp[0].copy(), allp[1].copy(), allp[2].copy(), allp[3].copy(), 0, allp.shape[1],
                                   r["lim"], bc)
                ok &= ref == int(tsum.item())
                report.append(dict(contract_step=extra, neighbors=ref, found=int(tsum.item()),
                                   focus_leaves=dom.view().num_focus_leaves, resorts=dom.view().resorts))
            x, y, z, h = [r[k][st:en].clone() for k in "xyzh"]
    if a.impl == "native":
        # a quiet stretch: the particles barely move any more, so the local order of a sync can be repaired from the
        # previous one (the incremental re-sort, csrc/resort.hpp; the first syncs behind the large moves above give up and
        # wait four syncs).  Same invariants; with periodic boundaries (the box cannot change) it must have happened.
        resorts0 = dom.view().resorts
        for extra in range(7):
            r = dom.sync(x, y, z, h)
            st, en = r["start"], r["end"]
            keys = r["keys"].cpu().numpy().view(kdt)
            tot = torch.tensor([en - st], dtype=torch.int64)
            dist.all_reduce(tot)
            v = dom.view()
            ok &= int(tot.item()) == N and bool(np.all(keys[1:] >= keys[:-1]))
            ok &= bool(np.all(keys[st:en] >= kdt(v.range_start))) and (en == st or int(keys[en - 1]) < v.range_end)
            lx, ly, lz, lh = [r[k].cpu().numpy() for k in "xyzh"]
            tsum = torch.tensor([neighbor_sum(o, lx, ly, lz, lh, st, en, r["lim"], bc)], dtype=torch.int64)
            dist.all_reduce(tsum)
            parts = [None] * P
            dist.all_gather_object(parts, np.stack([lx[st:en], ly[st:en], lz[st:en], lh[st:en]]))
            if rank == 0:
                allp = np.concatenate(parts, axis=1)
                ok &= neighbor_sum(o, allp[0].copy(), allp[1].copy(), allp[2].copy(), allp[3].copy(), 0, allp.shape[1],
                                   r["lim"], bc) == int(tsum.item())
            m = en - st
            drift = torch.from_numpy(rng.normal(0, 2e-5, (m, 3))).to(dev).to(rdt)
            x, y, z, h = [r[k][st:en].clone() for k in "xyzh"]
            x, y, z = x + drift[:, 0], y + drift[:, 1], z + drift[:, 2]
            if a.pbc:
                x, y, z = [torch.remainder(v_, 1.0).clamp_(0.0, top) for v_ in (x, y, z)]
            else:
                x, y, z = x.clamp(0.0, top), y.clamp(0.0, top), z.clamp(0.0, top)
        resorted = torch.tensor([dom.view().resorts - resorts0], dtype=torch.int64)
        dist.all_reduce(resorted, op=dist.ReduceOp.MIN)
        if a.pbc and os.environ.get("CSTONE_NO_RESORT") is None:
            ok &= int(resorted.item()) >= 1
        if rank == 0:
            report.append(dict(quiet_syncs=7, resorted_on_every_rank=int(resorted.item())))
        tag64 = x * 3.0 + y * 5.0 + z * 7.0 + h
        tag32 = (x + 2.0 * y).to(torch.float32)
    if a.impl == "native" and a.key_bits == 64:
        # particles flagged with the remove marker leave the domain: every 10th of what each rank holds
        marks = torch.zeros(x.numel(), dtype=torch.int64, device=x.device)
        marks[::10] = -(1 << 63)
        gone = torch.tensor([int((marks != 0).sum())], dtype=torch.int64)
        before = torch.tensor([x.numel()], dtype=torch.int64)
        late = late_fields(x, y, z, h)
        r = dom.sync(x, y, z, h, props=[tag64, tag32], keys=marks)
        ok &= check_reapply(dom, late, r)
        after = torch.tensor([r["end"] - r["start"]], dtype=torch.int64)
        for t in (gone, before, after):
            dist.all_reduce(t)
        ok &= int(after.item()) == int(before.item()) - int(gone.item())
        kk = r["keys"].cpu().numpy().view(np.uint64)[r["start"]:r["end"]]
        ok &= bool(np.all(kk < np.uint64(1 << 63))) and bool(np.all(kk[1:] >= kk[:-1]))
        ok &= bool(torch.equal(r["props"][0][r["start"]:r["end"]], tag64_of(r, r["start"], r["end"])))
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("DIST_RESULT " + json.dumps(dict(ok=bool(flag.item()), ranks=P, report=report)))
    dist.destroy_process_group()
    sys.exit(0 if flag.item() else 1)


if __name__ == "__main__":
    main()
