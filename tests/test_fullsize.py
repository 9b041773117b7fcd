"""BASELINE.json's full sizes on the GPU through size-independent properties (the oracle would take minutes there):
sortedness / permutation / stability of the sort, and for Domain::sync: keys sorted and consistent with the
coordinates they travel with, particle multiset preserved, leaf counts add up and respect the bucket size, idempotence."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))

pytestmark = pytest.mark.gpu


def _u64(t):
    return t.view(np.uint64) if isinstance(t, np.ndarray) else t


@pytest.mark.parametrize("kb", [32, 64])
def test_sort_1e8_pairs_sorted_stable_permutation(hip, kb):
    import torch

    n = 100_000_000
    g = torch.Generator(device="cuda").manual_seed(kb)
    if kb == 64:
        # 24 random bits spread over the low, middle and high bytes: every digit pass works and every key has ties
        r = torch.randint(0, 1 << 24, (n,), dtype=torch.int64, device="cuda", generator=g)
        keys = (r & 0xFF) | ((r & 0xFF00) << 24) | ((r & 0xFF0000) << 39)
    else:
        keys = torch.randint(0, 1 << 22, (n,), dtype=torch.int32, device="cuda", generator=g) << 8
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    src = keys.clone()
    hip.sort_pairs(keys, vals)
    hip.sync()
    assert bool((keys[1:] >= keys[:-1]).all())                      # sorted (keys are non-negative as signed, too)
    assert bool((src[vals.long()] == keys).all())                   # the payload is the permutation that sorts
    same = keys[1:] == keys[:-1]
    assert bool((vals[1:][same] > vals[:-1][same]).all())           # stable: ties keep their input order
    assert int(same.sum()) > n // 2
    seen = torch.zeros(n, dtype=torch.bool, device="cuda")
    seen[vals.long()] = True
    assert bool(seen.all())                                         # a permutation: every index exactly once


def test_domain_sync_1e8_uniform_invariants(hip):
    import torch

    import cstone_amd
    from cstone_amd.domain import Domain

    n, bucket_focus = 100_000_000, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    x, y, z = [torch.rand(n, dtype=torch.float64, device="cuda", generator=g) for _ in range(3)]
    h = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 1e-3 + 1e-4
    sums = [float(a.sum()) for a in (x, y, z, h)]
    ident = x * 3.0 + y * 5.0 + z * 7.0 + h  # travels as a property: must stay attached to its particle
    keys = torch.zeros(n, dtype=torch.int64, device="cuda")
    scratch = torch.empty(n, dtype=torch.float64, device="cuda")
    dom = Domain(hip, cstone_amd.HILBERT, 64, 64, n // 100, bucket_focus, 0.5, cstone_amd.make_cbox([0, 1] * 3))
    keys, x, y, z, h, scratch, (ident,) = dom.sync(keys, x, y, z, h, scratch, [ident])
    hip.sync()
    v = dom.view()
    assert (v.start_index, v.end_index, v.num_particles_with_halos) == (0, n, n)
    assert bool((keys[1:] >= keys[:-1]).all())
    # keys belong to the coordinates next to them
    again = hip.compute_sfc_keys(cstone_amd.HILBERT, 64, x, y, z, v.box)
    assert bool((again == keys).all())
    del again
    # the same particles, every field still attached
    assert bool((ident == x * 3.0 + y * 5.0 + z * 7.0 + h).all())
    for a, s0 in zip((x, y, z, h), sums):
        assert abs(float(a.sum()) - s0) <= 1e-9 * abs(s0)
    # focus tree: counts add up, respect the bucket, match the keys
    L = v.num_focus_leaves
    counts = dom.fetch(v.focus_leaf_counts, L, np.uint32)
    leaves = dom.fetch(v.focus_leaves, L + 1, np.uint64)
    layout = dom.fetch(v.layout, L + 1, np.uint32)
    assert int(counts.sum(dtype=np.uint64)) == n and counts.max() <= bucket_focus
    assert leaves[0] == 0 and leaves[-1] == 1 << 63 and np.all(leaves[1:] > leaves[:-1])
    assert np.array_equal(layout, np.concatenate([[0], np.cumsum(counts, dtype=np.uint64)]).astype(np.uint32))
    probe = np.linspace(0, L - 1, 2000).astype(np.int64)
    kq = torch.from_numpy(leaves[probe].view(np.int64)).cuda()
    pos = hip.lower_bound(keys, kq).cpu().numpy()
    assert np.array_equal(pos, layout[probe].astype(np.int64))
    # a second sync of the unchanged, already sorted particles changes nothing
    k2, x2, y2, z2, h2, scratch, _ = dom.sync(keys.clone(), x.clone(), y.clone(), z.clone(), h.clone(), scratch)
    hip.sync()
    assert bool((k2 == keys).all()) and bool((x2 == x).all()) and bool((z2 == z).all()) and bool((h2 == h).all())
    v2 = dom.view()
    assert v2.num_focus_leaves == L
    assert np.array_equal(dom.fetch(v2.focus_leaves, L + 1, np.uint64), leaves)


def test_neighbors_plummer_sample_against_oracle(hip, oracle):
    """2e6 Plummer particles, bucket 64: the lists of 3000 scattered targets equal the oracle's (bit-exact order)"""
    import torch

    from oracle.oracle import HILBERT, Box

    n, ngmax = 2_000_000, 160
    rng = np.random.default_rng(12)
    u = np.clip(rng.uniform(size=n), 1e-12, 1.0)
    r = np.minimum(1.0 / np.sqrt(np.maximum(u ** (-2.0 / 3.0) - 1.0, 1e-12)), 10.0)
    ct, ph = rng.uniform(-1, 1, n), rng.uniform(0, 2 * np.pi, n)
    st = np.sqrt(1 - ct * ct)
    x, y, z = r * st * np.cos(ph), r * st * np.sin(ph), r * ct
    rho = 3.0 * n / (4 * np.pi) * (1 + r * r) ** -2.5
    h = np.minimum(0.5 * (3.0 * 100 / (4 * np.pi * rho)) ** (1 / 3), 1.0)
    box = Box([-10.001, 10.001] * 3, (0, 0, 0))
    keys = oracle.compute_sfc_keys(HILBERT, 64, x, y, z, box)
    ks, order = oracle.sort_pairs(keys, np.arange(n))
    x, y, z, h = x[order], y[order], z[order], h[order]
    tree, counts = oracle.compute_octree(ks, 64)
    o = oracle.build_octree(tree)
    layout = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    cen, siz = oracle.node_centers(HILBERT, o["prefixes"], box, 64)

    import cstone_amd

    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    od = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in o.items()}
    cb = cstone_amd.make_cbox(box.lim, box.bc)
    xd, yd, zd, hd, ld, cd, sd = dev(x), dev(y), dev(z), dev(h), dev(layout), dev(cen), dev(siz)
    for first in (0, n // 2 - 500, n - 1000):  # centre of the sphere (dense), halo (sparse), both ends of the curve
        last = first + 1000
        n_ref, c_ref = oracle.find_neighbors(x, y, z, h, first, last, box, o, layout, cen, siz, ngmax)
        n_got, c_got = hip.find_neighbors(xd, yd, zd, hd, first, last, cb, od, ld, cd, sd, ngmax)
        hip.sync()
        c_got, n_got = c_got.cpu().numpy().view(np.uint32), n_got.cpu().numpy().view(np.uint32)
        assert np.array_equal(c_got, c_ref)
        mask = np.arange(ngmax)[None, :] < np.minimum(c_ref, ngmax)[:, None]
        assert np.array_equal(n_got[mask], n_ref[mask])


def test_domain_sync_1e8_plummer_with_neighbors(hip):
    """BASELINE configs[2]: 10^8 Plummer-sphere particles through Domain::sync (bucketSize 64: a deep, very uneven
    tree), then findNeighbors on the domain's own octree; the neighbour counts of scattered targets are checked against a
    brute-force pass over all 10^8 particles that evaluates the same IEEE operations in the same order"""
    import math

    import torch

    import cstone_amd
    from cstone_amd.domain import Domain

    n, bucket_focus = 100_000_000, 64
    # the reference's own cloud (test/coord_samples/plummer.hpp restated: srand48(42), R < 100, no clamping shell), h from
    # the local density; the domain measures its tight box itself
    from cstone_amd import clouds

    x, y, z, h, lim = clouds.make_cloud("plummer", n, n, "cuda", torch.float64, 0)
    assert float(x.abs().max()) > 30.0  # the halo of the sphere reaches out to R = 100 * 3 pi / 16
    ident = x * 3.0 + y * 5.0 + z * 7.0 + h
    keys = torch.zeros(n, dtype=torch.int64, device="cuda")
    scratch = torch.empty(n, dtype=torch.float64, device="cuda")
    box = cstone_amd.make_cbox(lim)
    dom = Domain(hip, cstone_amd.HILBERT, 64, 64, n // 100, bucket_focus, 0.5, box)
    for sync in range(2):  # the second call takes the steady-state path (partial radix passes + run fix-up)
        keys, x, y, z, h, scratch, (ident,) = dom.sync(keys, x, y, z, h, scratch, [ident])
        hip.sync()
        v = dom.view()
        assert (v.start_index, v.end_index, v.num_particles_with_halos) == (0, n, n)
        assert bool((keys[1:] >= keys[:-1]).all())
        assert bool((hip.compute_sfc_keys(cstone_amd.HILBERT, 64, x, y, z, v.box) == keys).all())
        assert bool((ident == x * 3.0 + y * 5.0 + z * 7.0 + h).all())
    L = v.num_focus_leaves
    counts = dom.fetch(v.focus_leaf_counts, L, np.uint32)
    leaves = dom.fetch(v.focus_leaves, L + 1, np.uint64)
    layout = dom.fetch(v.layout, L + 1, np.uint32)
    assert int(counts.sum(dtype=np.uint64)) == n and counts.max() <= bucket_focus
    assert leaves[0] == 0 and leaves[-1] == 1 << 63 and np.all(leaves[1:] > leaves[:-1])
    assert np.array_equal(layout, np.concatenate([[0], np.cumsum(counts, dtype=np.uint64)]).astype(np.uint32))
    level = 21 - (np.log2(np.diff(leaves).astype(np.float64)) / 3).round().astype(int)
    assert level.max() >= 10 and level.min() <= 4  # the core of the sphere needs a deep tree, the outskirts do not

    # neighbor search on the domain's octree view (device pointers wrapped as tensors)
    from cstone_amd.distributed import _DevMem

    def wrap(ptr, dt, count):
        return torch.as_tensor(_DevMem(ptr, count * torch.empty(0, dtype=dt).element_size()), device="cuda").view(dt)

    M = v.num_focus_nodes
    oc = dict(child_offsets=wrap(v.child_offsets, torch.int32, M + 1),
              internal_to_leaf=wrap(v.internal_to_leaf, torch.int32, M))
    lay = wrap(v.layout, torch.int32, L + 1)
    cen, siz = wrap(v.centers, torch.float64, 3 * M), wrap(v.sizes, torch.float64, 3 * M)
    rng = np.random.default_rng(3)
    for first in (0, n // 2, int(rng.integers(0, n - 70000)), n - 65536):
        last = first + 65536
        _, nc = hip.find_neighbors(x, y, z, h, first, last, v.box, oc, lay, cen, siz, 0)
        hip.sync()
        assert 20 < float(nc.double().mean()) < 2000  # about 100 by construction
        for i in rng.integers(first, last, 6):
            i = int(i)
            dx, dy, dz = x - x[i], y - y[i], z - z[i]
            d2 = dx * dx + dy * dy + dz * dz
            brute = int((d2 < 4.0 * h[i] * h[i]).sum()) - 1
            assert int(nc[i - first]) == brute, (i, int(nc[i - first]), brute)


def test_domain_sync_clustered_2p5e8(hip):
    """twice the per-GPU share of BASELINE configs[4] (10^9 clustered particles on 8 GPUs) on one GPU: half the
    particles in a uniform background, half in a few tight blobs; sizes beyond 2^27 exercise the 32-bit index arithmetic
    of the sort tiles and the leaf layout"""
    import torch

    import cstone_amd
    from cstone_amd.domain import Domain

    n, bucket_focus = 250_000_000, 64
    g = torch.Generator(device="cuda").manual_seed(23)
    centers = torch.rand((6, 3), dtype=torch.float64, device="cuda", generator=g) * 0.6 + 0.2
    which = torch.randint(0, 12, (n,), device="cuda", generator=g)  # 0..5: a blob, 6..11: background
    cols = []
    for d in range(3):
        v = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
        blob = centers[which.clamp(max=5), d] + 0.01 * torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
        cols.append(torch.where(which < 6, blob, v).clamp_(0.0, 1.0))
        del v, blob
    x, y, z = cols
    del cols, which
    h = torch.full((n,), 1e-3, dtype=torch.float64, device="cuda")
    sx = float(x.sum())
    keys = torch.zeros(n, dtype=torch.int64, device="cuda")
    scratch = torch.empty(n, dtype=torch.float64, device="cuda")
    dom = Domain(hip, cstone_amd.HILBERT, 64, 64, n // 100, bucket_focus, 0.5, cstone_amd.make_cbox([0, 1] * 3))
    for sync in range(2):
        keys, x, y, z, h, scratch, _ = dom.sync(keys, x, y, z, h, scratch)
        hip.sync()
        v = dom.view()
        assert (v.start_index, v.end_index, v.num_particles_with_halos) == (0, n, n)
        assert bool((keys[1:] >= keys[:-1]).all())
        assert bool((hip.compute_sfc_keys(cstone_amd.HILBERT, 64, x, y, z, v.box) == keys).all())
        assert abs(float(x.sum()) - sx) <= 1e-9 * sx
    L = v.num_focus_leaves
    counts = dom.fetch(v.focus_leaf_counts, L, np.uint32)
    layout = dom.fetch(v.layout, L + 1, np.uint32)
    assert int(counts.sum(dtype=np.uint64)) == n and counts.max() <= bucket_focus and int(layout[-1]) == n


@pytest.mark.parametrize("dist_name,n,n_global", [("uniform", 12_500_000, 100_000_000),
                                                   ("clustered", 125_000_000, 1_000_000_000)],
                         ids=["configs3-share-1.25e7-uniform", "configs4-share-1.25e8-clustered"])
def test_one_ranks_share_through_the_multi_rank_sync(hip, dist_name, n, n_global):
    """BASELINE configs[3] / configs[4] are 8-GPU runs: ONE rank's share of each (1.25e7 uniform, 1.25e8 clustered
    particles; h as in the whole cloud) goes through cstone_hip_domain_mr_sync -- the code path of the N-GPU run, RCCL
    collectives with a communicator of one rank -- over three syncs with every particle drifting: keys sorted and
    consistent with the coordinates, a property still attached, the focus tree's counts / layout / bucket bound, and
    neighbour counts of a sample against a brute-force pass"""
    import socket

    import torch
    import torch.distributed as dist

    import cstone_amd
    from cstone_amd import clouds
    from cstone_amd.distributed import NativeDistributedDomain, RcclCollectives

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        coll = RcclCollectives(hip)
        assert coll.size == 1
        x, y, z, h, lim = clouds.make_cloud(dist_name, n, n_global, "cuda", torch.float64, 42, 3, 8)
        bucket_focus = 64
        dom = NativeDistributedDomain(hip, cstone_amd.HILBERT, 64, 64, max(64, n_global // 800), bucket_focus, lim,
                                      (0, 0, 0), coll=coll)
        g = torch.Generator(device="cuda").manual_seed(9)
        for sync in range(3):
            ident = x * 3.0 + y * 5.0 + z * 7.0 + h
            r = dom.sync(x, y, z, h, props=[ident])
            hip.sync()
            st, en = r["start"], r["end"]
            assert (st, en) == (0, n) and r["x"].numel() == n
            keys = r["keys"]
            assert bool((keys[1:] >= keys[:-1]).all())
            v = dom.view()
            assert bool((hip.compute_sfc_keys(cstone_amd.HILBERT, 64, r["x"], r["y"], r["z"], v.box) == keys).all())
            assert bool((r["props"][0] == r["x"] * 3.0 + r["y"] * 5.0 + r["z"] * 7.0 + r["h"]).all())
            L = v.num_focus_leaves
            counts = dom.fetch(v.focus_leaf_counts, L, np.uint32)
            layout = dom.fetch(v.layout, L + 1, np.uint32)
            assert int(counts.sum(dtype=np.uint64)) == n and int(layout[-1]) == n
            # (the converged tree of the first sync respects the bucket; afterwards a sync takes ONE update step, like the
            #  reference: a leaf that particles drifted into may exceed the bucket until the next sync splits it)
            if sync == 0:
                assert counts.max() <= bucket_focus
            assert np.array_equal(np.diff(layout.astype(np.int64)), counts.astype(np.int64))
            assert (v.start_cell, v.end_cell) == (0, L)
            if sync == 2:
                oc = dom.octree()
                rng = np.random.default_rng(5)
                first = int(rng.integers(0, n - 70000))
                _, nc = hip.find_neighbors(r["x"], r["y"], r["z"], r["h"], first, first + 65536, v.box, oc, oc["layout"],
                                           oc["centers"], oc["sizes"], 0)
                hip.sync()
                for i in rng.integers(first, first + 65536, 5):
                    i = int(i)
                    dx, dy, dz = r["x"] - r["x"][i], r["y"] - r["y"][i], r["z"] - r["z"][i]
                    d2 = dx * dx + dy * dy + dz * dz
                    assert int(nc[i - first]) == int((d2 < 4.0 * r["h"][i] * r["h"][i]).sum()) - 1
                    del dx, dy, dz, d2
            # the client's time step: every particle by up to 0.1 h per coordinate
            x, y, z, h = [r[k][st:en] for k in "xyzh"]
            for a in (x, y, z):
                d = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
                a.add_(d.sub_(0.5).mul_(0.2).mul_(h)).clamp_(0.0, 1.0)
                del d
        # the steady state re-sorts from the previous order (the clustered cloud piles its clamped tails up on the faces of
        # the box: leaves of hundreds of equal keys, beyond what the leaf pass takes -- it keeps the radix path)
        if dist_name == "uniform":
            assert int(dom.view().resorts) >= 1
        dom.close()
        coll.close()
    finally:
        dist.destroy_process_group()


def test_tear_down_in_the_wrong_order_is_an_error_code_not_a_crash():
    """gpurun_out/r3_let1.log: a client that destroyed its context first got a SIGSEGV from the buffers it released
    afterwards.  The library keeps a registry of live contexts: free and the destroy functions on a dead context return
    CSTONE_E_ARG (and still release what they own)"""
    import ctypes as C

    import torch

    import cstone_amd
    from cstone_amd.domain import Domain

    ctx = cstone_amd.Context(0)
    lib, raw = ctx.lib, C.c_void_p(ctx.h.value if hasattr(ctx.h, "value") else ctx.h)
    buf = C.c_void_p()
    assert lib.cstone_hip_malloc(raw, C.byref(buf), C.c_size_t(1 << 20)) == 0
    n = 50_000
    g = torch.Generator(device="cuda").manual_seed(1)
    x, y, z = [torch.rand(n, dtype=torch.float64, device="cuda", generator=g) for _ in range(3)]
    h = torch.full((n,), 0.01, dtype=torch.float64, device="cuda")
    dom = Domain(ctx, cstone_amd.HILBERT, 64, 64, 64, 16, 0.5, cstone_amd.make_cbox([0, 1] * 3))
    dom.sync(torch.zeros(n, dtype=torch.int64, device="cuda"), x, y, z, h, torch.empty_like(x))
    ctx.sync()
    dom_handle = C.c_void_p(dom.h.value)
    dom.h = None  # (the wrapper must not destroy it again)
    assert lib.cstone_hip_ctx_destroy(raw) == 0
    ctx.h = None
    assert lib.cstone_hip_free(raw, buf) == -1           # CSTONE_E_ARG; the buffer is released all the same
    assert lib.cstone_hip_domain_destroy(dom_handle) == -1
    assert lib.cstone_hip_ctx_destroy(raw) == -1         # a second destroy of the same pointer


def test_encode_sort_tree_1e7_against_reference_digests(hip, oracle):
    """BASELINE configs[1] (10^7 uniform particles: encode + radix sort + cornerstone tree, no halos) BIT FOR BIT against
    the reference: tests/golden/ref_1e7_digests.json holds the SHA-256 of what the reference's own computeSfcKeys,
    sort_by_key and computeOctree produce on its RandomCoordinates cloud (tests/golden/make_golden_1e7.py, run where
    /root/reference exists); the input is regenerated here by the restated generator"""
    import hashlib
    import json

    import torch

    import cstone_amd
    from oracle import oracle as orc

    want = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_1e7_digests.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    n, box = want["n"], orc.Box(want["box"])
    x, y, z = oracle.random_uniform(n, box, want["seed"])
    assert sha(x) == want["x_sha256"]  # the same cloud as the one the reference saw
    cb = cstone_amd.make_cbox(box.lim, box.bc)
    xd, yd, zd = [torch.from_numpy(a).cuda() for a in (x, y, z)]
    keys = hip.compute_sfc_keys(cstone_amd.HILBERT, 64, xd, yd, zd, cb)
    assert sha(cstone_amd.keys_to_numpy(keys, 64)) == want["keys_sha256"]
    order = torch.arange(n, dtype=torch.int32, device="cuda")
    hip.sort_pairs(keys, order)
    hip.sync()
    assert sha(cstone_amd.keys_to_numpy(keys, 64)) == want["sorted_keys_sha256"]
    assert sha(order.cpu().numpy().view(np.uint32)) == want["order_sha256"]
    tree, counts, _ = hip.compute_octree(keys, want["bucket"])
    hip.sync()
    assert tree.numel() - 1 == want["num_leaves"] == 262256
    assert sha(cstone_amd.keys_to_numpy(tree, 64)) == want["leaves_sha256"]
    assert sha(counts.cpu().numpy().view(np.uint32)) == want["counts_sha256"]
    # the fused entry (encode + digit counting + sort, what Domain::sync issues) gives the same keys and ordering
    k2, o2 = hip.sfc_keys_and_ordering(cstone_amd.HILBERT, 64, xd, yd, zd, cb)
    hip.sync()
    assert sha(cstone_amd.keys_to_numpy(k2, 64)) == want["sorted_keys_sha256"]
    assert sha(o2.cpu().numpy().view(np.uint32)) == want["order_sha256"]


def test_resort_full_size_drift(hip):
    """The mover path of the incremental re-sort (csrc/resort.hpp) at BASELINE's full size: 10^8 uniform particles through
    the bench's time-stepping loop -- three steps in which EVERY particle drifts by up to 0.1 h per coordinate (7 % of
    them leave their leaf) and one in which 1 % jump by up to 2h -- once through a domain that may re-sort and once through
    one that may not (cstone_hip_domain_set_sort_mode): keys, coordinates, h, the particle identities, layout and leaf array are equal
    after every sync, the keys are the encode of the coordinates next to them, and all four syncs were re-sorted."""
    import torch

    import cstone_amd
    from cstone_amd.domain import Domain

    n, bucket_focus = 100_000_000, 64
    h0 = 0.6 * (3.0 * 100 / (4 * np.pi * n)) ** (1.0 / 3.0)

    def fresh():
        g = torch.Generator(device="cuda").manual_seed(77)
        x, y, z = [torch.rand(n, dtype=torch.float64, device="cuda", generator=g) for _ in range(3)]
        h = torch.full((n,), h0, dtype=torch.float64, device="cuda")
        ident = torch.arange(n, dtype=torch.float64, device="cuda")
        dom = Domain(hip, cstone_amd.HILBERT, 64, 64, n // 100, bucket_focus, 0.5, cstone_amd.make_cbox([0, 1] * 3))
        return dict(dom=dom, k=torch.zeros(n, dtype=torch.int64, device="cuda"), x=x, y=y, z=z, h=h, id=ident,
                    s=torch.empty_like(x))

    def sync(d, allow):
        d["dom"].set_sort_mode(d["dom"].SORT_INCREMENTAL if allow else d["dom"].SORT_FROM_SCRATCH)
        d["k"], d["x"], d["y"], d["z"], d["h"], d["s"], (d["id"],) = d["dom"].sync(d["k"], d["x"], d["y"], d["z"],
                                                                                 d["h"], d["s"], [d["id"]])
        hip.sync()

    def move(d, kind, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        if kind == "drift":
            for a in (d["x"], d["y"], d["z"]):
                t = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
                a.add_(t.sub_(0.5).mul_(0.2 * h0)).clamp_(0.0, 1.0)
                del t
        else:
            # (distinct indices: an indexed assignment with repeated indices keeps an arbitrary one of the writes, the two
            #  domains of this test would drift apart)
            idx = torch.unique(torch.randint(0, n, (n // 100,), device="cuda", generator=g))
            m = idx.numel()
            for a in (d["x"], d["y"], d["z"]):
                t = (torch.rand(m, dtype=torch.float64, device="cuda", generator=g) - 0.5) * (4 * h0)
                a[idx] = (a[idx] + t).clamp_(0.0, 1.0)

    a, b = fresh(), fresh()
    sync(a, True), sync(b, False)
    for step, kind in enumerate(["drift", "drift", "jump", "drift"]):
        move(a, kind, 100 + step), move(b, kind, 100 + step)
        sync(a, True), sync(b, False)
        va, vb = a["dom"].view(), b["dom"].view()
        assert (va.end_index, va.num_focus_leaves) == (vb.end_index, vb.num_focus_leaves) == (n, va.num_focus_leaves), step
        for f in ("k", "x", "y", "z", "h", "id"):
            assert bool(torch.equal(a[f], b[f])), (step, kind, f)
        L = va.num_focus_leaves
        assert np.array_equal(a["dom"].fetch(va.layout, L + 1, np.uint32), b["dom"].fetch(vb.layout, L + 1, np.uint32)), step
        assert np.array_equal(a["dom"].fetch(va.focus_leaves, L + 1, np.uint64),
                              b["dom"].fetch(vb.focus_leaves, L + 1, np.uint64)), step
        again = hip.compute_sfc_keys(cstone_amd.HILBERT, 64, a["x"], a["y"], a["z"], va.box)
        assert bool((again == a["k"]).all()) and bool((a["k"][1:] >= a["k"][:-1]).all()), (step, kind)
        del again
    sa, sb = a["dom"].stats(), b["dom"].stats()
    # (the first drift step moves the outermost particles: the open box follows once, that sync encodes again and sorts
    #  from scratch; every other sync is re-sorted)
    assert sa["resorts"] + sa["box_redos"] == 4 and sa["resorts"] >= 3 and sa["resort_fallbacks"] == 0, sa
    assert sb["resorts"] == 0, sb
    assert sa["last_movers"] > n // 50, sa  # the drift step: several per cent of the particles changed their leaf
