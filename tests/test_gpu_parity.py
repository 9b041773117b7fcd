"""GPU parity tests (run with -m gpu on an MI355X): every entry point of the C ABI against the CPU oracle on the
same seeded inputs.  Integer / byte / index results must be bit-exact; neighbor lists are compared exactly too
(the HIP path evaluates the same IEEE operations in the same order as the reference's CPU path, no FMA)."""
import numpy as np
import pytest

from helpers import Box, OctreeMaker, end_key, key_dtype, max_level, random_cloud, real_dtype
from oracle.oracle import HILBERT, MORTON

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    return torch


def dev(a):
    """numpy -> device tensor (unsigned arrays travel as the signed type of the same width)"""
    torch = _torch()
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    elif a.dtype == np.uint64:
        a = a.view(np.int64)
    return torch.from_numpy(a.copy()).cuda()


def host(t, unsigned=True):
    a = t.cpu().numpy()
    if unsigned and a.dtype == np.int32:
        return a.view(np.uint32)
    if unsigned and a.dtype == np.int64:
        return a.view(np.uint64)
    return a


def cbox(box):
    import cstone_amd

    return cstone_amd.make_cbox(box.lim, box.bc)


BOXES = [Box([0, 1]), Box([-1.3, 2.1, 0.2, 0.9, -5, 7])]


@pytest.mark.parametrize("kb", [32, 64])
@pytest.mark.parametrize("rb", [32, 64])
@pytest.mark.parametrize("curve", [MORTON, HILBERT])
def test_sfc_keys(hip, oracle, kb, rb, curve):
    for box in BOXES:
        for n in (1, 7, 1000, 100003):
            x, y, z = random_cloud(n, box, rb, seed=n + kb + rb)
            # box corners and faces are the clamping edge cases (sfc.hpp:166-168)
            lim = box.lim.astype(real_dtype(rb))
            x[0], y[0], z[0] = lim[1], lim[3], lim[5]
            if n > 2:
                x[1], y[1], z[1] = lim[0], lim[2], lim[4]
            ref = oracle.compute_sfc_keys(curve, kb, x, y, z, box)
            got = hip.compute_sfc_keys(curve, kb, dev(x), dev(y), dev(z), cbox(box))
            assert np.array_equal(host(got), ref), (kb, rb, curve, n)


@pytest.mark.parametrize("kb", [32, 64])
def test_sfc_keys_keep_remove_marker_and_unaligned(hip, oracle, kb):
    box = Box([0, 1])
    n = 5000
    x, y, z = random_cloud(n, box, 64, seed=3)
    keys = np.zeros(n, dtype=key_dtype(kb))
    keys[::7] = end_key(kb)  # particles flagged for removal keep their marker (sfc.hpp:289)
    ref = oracle.compute_sfc_keys(HILBERT, kb, x, y, z, box, keys.copy())
    got = hip.compute_sfc_keys(HILBERT, kb, dev(x), dev(y), dev(z), cbox(box), dev(keys))
    assert np.array_equal(host(got), ref)
    assert (ref[::7] == end_key(kb)).all()
    # sub-range views that are only 8-byte aligned take the scalar path (Domain::setupHalos encodes tails)
    xd, yd, zd, kd = dev(x), dev(y), dev(z), dev(np.zeros(n, dtype=key_dtype(kb)))
    hip.compute_sfc_keys(HILBERT, kb, xd[1:], yd[1:], zd[1:], cbox(box), kd[1:])
    ref2 = oracle.compute_sfc_keys(HILBERT, kb, x[1:], y[1:], z[1:], box)
    assert np.array_equal(host(kd)[1:], ref2)


@pytest.mark.parametrize("kb", [32, 64])
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 4095, 4096, 4097, 50000, 300001])
def test_sort_pairs_stable(hip, oracle, kb, n):
    torch = _torch()
    rng = np.random.default_rng(n + kb)
    top = end_key(kb)
    keys = rng.integers(0, top, n, dtype=np.uint64).astype(key_dtype(kb))
    if n > 10:
        keys[rng.integers(0, n, n // 3)] = keys[0]  # plenty of duplicates: stability is observable
        keys[-1] = key_dtype(kb)(np.iinfo(key_dtype(kb)).max)  # all bits set must sort last
    vals = np.arange(n, dtype=np.uint32)
    rk, rv = oracle.sort_pairs(keys, vals)
    kd, vd = dev(keys), dev(vals)
    hip.sort_pairs(kd, vd)
    hip.sync()
    assert np.array_equal(host(kd), rk)
    assert np.array_equal(host(vd), rv)
    # caller-provided scratch (the GpuSfcSorter calling convention, gather.cuh:81-98)
    if n:
        kd, vd = dev(keys), dev(vals)
        ka, va = torch.empty_like(kd), torch.empty_like(vd)
        tmp = torch.empty(hip.sort_temp_bytes(kb, n), dtype=torch.uint8, device="cuda")
        hip.sort_pairs(kd, vd, ka, va, tmp)
        hip.sync()
        assert np.array_equal(host(kd), rk) and np.array_equal(host(vd), rv)


@pytest.mark.parametrize("kb", [32, 64])
def test_sort_sorted_and_constant_inputs(hip, oracle, kb):
    n = 200000
    keys = np.sort(np.random.default_rng(1).integers(0, end_key(kb), n, dtype=np.uint64)).astype(key_dtype(kb))
    for k in (keys, keys[::-1].copy(), np.full(n, 12345, dtype=key_dtype(kb))):
        rk, rv = oracle.sort_pairs(k, np.arange(n))
        kd, vd = dev(k), dev(np.arange(n, dtype=np.uint32))
        hip.sort_pairs(kd, vd)
        assert np.array_equal(host(kd), rk) and np.array_equal(host(vd), rv)


def test_sort_large_properties(hip):
    """BASELINE config 2 size (10^7): sortedness + permutation checksum (size-independent properties)"""
    torch = _torch()
    n = 10_000_000
    g = torch.Generator(device="cuda").manual_seed(5)
    keys = torch.randint(0, 2**62, (n,), dtype=torch.int64, device="cuda", generator=g)
    orig = keys.clone()
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    hip.sort_pairs(keys, vals)
    hip.sync()
    assert bool((keys[1:] >= keys[:-1]).all())
    assert bool((orig[vals.long()] == keys).all())           # values carry the keys' origin
    assert int(vals.long().sum()) == n * (n - 1) // 2          # a permutation
    same = keys[1:] == keys[:-1]
    assert bool((vals[1:][same] > vals[:-1][same]).all())     # stability on ties


@pytest.mark.parametrize("eb", [1, 2, 4, 8, 12, 16, 24, 32])
def test_gather_scatter(hip, eb):
    n = 70001
    rng = np.random.default_rng(eb)
    src = rng.integers(0, 255, (n, eb), dtype=np.uint8)
    perm = rng.permutation(n).astype(np.uint32)
    sd, pd = dev(src), dev(perm)
    dd = _torch().zeros_like(sd)
    hip.gather(pd, sd, dd, elem_bytes=eb)
    assert np.array_equal(host(dd, False), src[perm])
    dd2 = _torch().zeros_like(sd)
    hip.scatter(pd, sd, dd2, elem_bytes=eb)
    ref = np.zeros_like(src)
    ref[perm] = src
    assert np.array_equal(host(dd2, False), ref)


@pytest.mark.parametrize("rb", [32, 64])
def test_minmax_and_scans(hip, rb):
    rng = np.random.default_rng(rb)
    for n in (1, 100, 100000, 1234567):
        x = rng.normal(0, 10, n).astype(real_dtype(rb))
        lo, hi = hip.minmax(dev(x))
        assert lo == x.min() and hi == x.max()
    # unaligned starts (the kernel reads 16-byte packs) with the extremes in the scalar head / tail
    x = rng.normal(0, 10, 300007).astype(real_dtype(rb))
    xd = dev(x)
    for off in (1, 2, 3):
        sub = x[off:off + 300000].copy()
        sub[0], sub[-1] = 1e6, -1e6
        xd[off:off + 300000].copy_(dev(sub))
        lo, hi = hip.minmax(xd[off:off + 300000])
        assert lo == -1e6 and hi == 1e6
    for n in (1, 255, 2048, 2049, 100000, 3000001):
        v = rng.integers(0, 100, n, dtype=np.uint32)
        vd = dev(v)
        out = _torch().zeros_like(vd)
        hip.exclusive_scan(vd, out, init=5)
        ref = np.concatenate([[0], np.cumsum(v, dtype=np.uint64)[:-1]]).astype(np.uint32) + 5
        assert np.array_equal(host(out), ref)
        offs = _torch().full((n + 1,), 77, dtype=vd.dtype, device=vd.device)
        hip.offsets_from_counts(vd, offs)  # the exclusive scan with its total behind it
        assert np.array_equal(host(offs), np.concatenate([[0], np.cumsum(v, dtype=np.uint64)]).astype(np.uint32))
        hip.inclusive_scan(vd, vd)  # in place
        assert np.array_equal(host(vd), np.cumsum(v, dtype=np.uint64).astype(np.uint32))
    # no counts: the one offset is zero
    offs = _torch().full((1,), 77, dtype=_torch().int32, device="cuda")
    hip.offsets_from_counts(offs[:0], offs)
    assert int(offs[0]) == 0
    # three small tables into one array (the read-back at the end of the halo layout)
    a, b, c = (rng.integers(0, 1 << 31, m, dtype=np.uint32) for m in (50, 900, 23))
    for n_a, n_b, use_a in ((7, 300, True), (7, 300, False), (0, 1, True), (300, 0, True)):
        imap = np.concatenate([rng.integers(0, a.size, n_a), rng.integers(0, b.size, n_b)]).astype(np.uint32)
        out = _torch().zeros(n_a + n_b + c.size, dtype=dev(a).dtype, device="cuda")
        hip.gather_tables(dev(imap), dev(a) if use_a else None, n_a, dev(b), n_b, dev(c), out)
        ref = np.concatenate([a[imap[:n_a]] if use_a else np.zeros(n_a, np.uint32), b[imap[n_a:]], c])
        assert np.array_equal(host(out).astype(np.uint32), ref)


def _sorted_keys(oracle, curve, kb, n, box, rb, seed, kind):
    x, y, z = random_cloud(n, box, rb, seed, kind)
    keys = oracle.compute_sfc_keys(curve, kb, x, y, z, box)
    ks, order = oracle.sort_pairs(keys, np.arange(n))
    return x[order], y[order], z[order], ks


@pytest.mark.parametrize("kb", [32, 64])
@pytest.mark.parametrize("kind", ["uniform", "gaussian", "clustered"])
@pytest.mark.parametrize("bucket", [1, 16, 64])
def test_octree_build_stepwise_and_converged(hip, oracle, kb, kind, bucket):
    n = 30000 if bucket > 1 else 3000
    _, _, _, keys = _sorted_keys(oracle, HILBERT, kb, n, Box([0, 1]), 64, seed=11, kind=kind)
    kd = dev(keys)
    # converged tree
    rt, rc = oracle.compute_octree(keys, bucket)
    gt, gc, iters = hip.compute_octree(kd, bucket)
    assert np.array_equal(host(gt), rt) and np.array_equal(host(gc), rc)
    # iteration-for-iteration parity from the root (Domain performs exactly one update per sync)
    tree = np.array([0, end_key(kb)], dtype=key_dtype(kb))
    counts = np.array([n], dtype=np.uint32)
    cap = rt.size + 4096 * 8
    tb = _torch().zeros(cap + 1, dtype=kd.dtype, device="cuda")
    cb = _torch().zeros(cap, dtype=_torch().int32, device="cuda")
    tb[:2] = dev(tree)
    cb[:1] = dev(counts)
    nl = 1
    for _ in range(iters + 1):
        # individual seam functions
        ops_ref, conv_ref = oracle.node_ops(tree, counts, bucket)
        ops_scan, new_n, conv = hip.compute_node_ops(tb, cb, bucket, num_nodes=nl)
        ex = np.concatenate([[0], np.cumsum(ops_ref)]).astype(np.int32)
        assert np.array_equal(host(ops_scan, False), ex) and new_n == ex[-1] and conv == conv_ref
        nt = hip.rebalance_tree(tb, ops_scan, new_n, num_nodes=nl)
        tree, counts, conv_o = oracle.update_octree(keys, bucket, tree, counts)
        assert np.array_equal(host(nt), tree)
        assert np.array_equal(host(hip.compute_node_counts(nt, kd)), counts)
        # composite update on capacity buffers
        nl, conv_h = hip.update_octree(kd, bucket, tb, cb, nl)
        assert conv_h == conv_o and nl == tree.size - 1
        assert np.array_equal(host(tb[:nl + 1]), tree) and np.array_equal(host(cb[:nl]), counts)
    assert conv_o


@pytest.mark.parametrize("kb", [32, 64])
def test_node_counts_max_count_and_empty(hip, oracle, kb):
    tree = OctreeMaker(kb).divide().divide(0).make()
    keys = np.sort(np.random.default_rng(2).integers(0, end_key(kb), 5000, dtype=np.uint64)).astype(key_dtype(kb))
    for mc in (0xFFFFFFFF, 100, 1):
        assert np.array_equal(host(hip.compute_node_counts(dev(tree), dev(keys), max_count=mc)),
                              oracle.node_counts(tree, keys, mc))
    # keys only in the last node / no keys at all
    last = np.full(10, end_key(kb) - 1, dtype=key_dtype(kb))
    assert np.array_equal(host(hip.compute_node_counts(dev(tree), dev(last))), oracle.node_counts(tree, last))
    empty = _torch().zeros(0, dtype=dev(tree).dtype, device="cuda")
    assert host(hip.compute_node_counts(dev(tree), empty)).sum() == 0


def test_update_octree_capacity_error(hip, oracle):
    keys = np.sort(np.random.default_rng(4).integers(0, end_key(64), 100000, dtype=np.uint64))
    tb = _torch().zeros(65, dtype=_torch().int64, device="cuda")
    cb = _torch().zeros(64, dtype=_torch().int32, device="cuda")
    tb[:2] = dev(np.array([0, end_key(64)], dtype=np.uint64))
    cb[:1] = 100000
    nl, _ = hip.update_octree(dev(keys), 16, tb, cb, 1)
    assert nl == -4096  # needs 4096 leaves, buffers untouched
    assert host(tb[:2]).tolist() == [0, end_key(64)]


def _octree_equal(got, ref):
    for k in ("prefixes", "child_offsets", "parents", "level_range", "internal_to_leaf", "leaf_to_internal"):
        g = host(got[k], unsigned=(k == "prefixes"))
        r = ref[k]
        if k == "parents":
            g = g[:r.size]
        if k == "child_offsets":
            g, r = g[:ref["num_nodes"]], r[:ref["num_nodes"]]
        assert np.array_equal(g, r), k


@pytest.mark.parametrize("kb", [32, 64])
def test_linked_octree(hip, oracle, kb):
    trees = [OctreeMaker(kb).make(), OctreeMaker(kb).divide().make(), OctreeMaker(kb).divide().divide(0).make(),
             OctreeMaker(kb).divide().divide(0).divide(0, 2).divide(3).make()]
    m = OctreeMaker(kb).divide()
    for i in range(8):
        m.divide(i)
    trees.append(m.make())
    cs = np.array([0, 1, 0o30173, 0o3333333333, end_key(kb) - 1, end_key(kb)], dtype=key_dtype(kb))
    trees.append(oracle.spanning_tree(cs))  # max-depth tree (octree.cpp:202-216)
    for kind in ("uniform", "clustered"):
        _, _, _, keys = _sorted_keys(oracle, HILBERT, kb, 40000, Box([0, 1]), 64, seed=21, kind=kind)
        trees.append(oracle.compute_octree(keys, 8)[0])
    for t in trees:
        ref = oracle.build_octree(t)
        got = hip.build_octree(dev(t))
        _octree_equal(got, ref)
        # saturating upsweep of leaf counts
        rng = np.random.default_rng(t.size)
        lc = rng.integers(0, 2**31, t.size - 1, dtype=np.uint32)
        q_ref = oracle.upsweep_counts(ref, lc)
        q = np.zeros(ref["num_nodes"], dtype=np.uint32)
        q[ref["leaf_to_internal"][ref["num_internal"]:]] = lc
        qd = dev(q)
        hip.upsweep_sum(got, qd)
        assert np.array_equal(host(qd), q_ref)


@pytest.mark.parametrize("kb", [32, 64])
@pytest.mark.parametrize("rb", [32, 64])
@pytest.mark.parametrize("curve", [MORTON, HILBERT])
def test_node_centers(hip, oracle, kb, rb, curve):
    box = BOXES[1]
    _, _, _, keys = _sorted_keys(oracle, curve, kb, 20000, box, rb, seed=5, kind="gaussian")
    tree, _ = oracle.compute_octree(keys, 16)
    o = oracle.build_octree(tree)
    rc, rs = oracle.node_centers(curve, o["prefixes"], box, rb)
    gc, gs = hip.node_centers(curve, dev(o["prefixes"]), cbox(box), rb)
    assert np.array_equal(host(gc, False), rc) and np.array_equal(host(gs, False), rs)


@pytest.mark.parametrize("kb", [32, 64])
@pytest.mark.parametrize("curve", [MORTON, HILBERT])
@pytest.mark.parametrize("bc", [(0, 0, 0), (1, 1, 1), (1, 0, 1)])
def test_find_halos(hip, oracle, kb, curve, bc):
    box = Box([-1.3, 2.1, 0.2, 0.9, -5, 7], bc)
    for kind, n, bucket in (("uniform", 60000, 16), ("clustered", 40000, 8)):
        x, y, z, keys = _sorted_keys(oracle, curve, kb, n, box, 64, seed=31, kind=kind)
        tree, counts = oracle.compute_octree(keys, bucket)
        o = oracle.build_octree(tree)
        od = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in o.items()}
        nl = tree.size - 1
        rng = np.random.default_rng(nl)
        for first, last in ((0, nl // 4), (nl // 4, 3 * nl // 4), (nl - 17, nl), (0, nl), (5, 5)):
            # per-leaf radii from smoothing lengths, like Halos::discover
            h = (rng.uniform(0.2, 1.5, n) * 0.02).astype(np.float64)
            layout = np.concatenate([[0], np.cumsum(counts[first:last])]).astype(np.uint32)
            hl = h[int(counts[:first].sum()):]
            r_ref = oracle.halo_radii(hl, layout, first, last, nl, 1.0)
            r_got = hip.halo_radii(dev(hl), dev(layout), first, last, nl, 1.0)
            assert np.array_equal(host(r_got, False), r_ref)
            f_ref = oracle.find_halos(curve, o, tree, r_ref, box, first, last)
            f_got = hip.find_halos(curve, od, dev(tree), r_got, cbox(box), first, last)
            hip.sync()
            assert np.array_equal(host(f_got, False), f_ref), (kind, first, last, f_ref.sum())
        # a large radius reaches across the whole box (deep stacks, PBC wrap on both sides)
        big = np.full(nl, 0.4 * (box.lim[3] - box.lim[2]), dtype=np.float32)
        f_ref = oracle.find_halos(curve, o, tree, big, box, nl // 3, nl // 2)
        f_got = hip.find_halos(curve, od, dev(tree), dev(big), cbox(box), nl // 3, nl // 2)
        hip.sync()
        assert np.array_equal(host(f_got, False), f_ref)


@pytest.mark.parametrize("rb", [32, 64])
@pytest.mark.parametrize("bc", [(0, 0, 0), (1, 1, 1), (0, 1, 0)])
def test_find_neighbors(hip, oracle, rb, bc):
    box = Box([0, 1, 0, 1, 0, 1], bc)
    n, ngmax = 20000, 64
    for kind in ("uniform", "clustered"):
        x, y, z, keys = _sorted_keys(oracle, HILBERT, 64, n, box, rb, seed=41, kind=kind)
        rng = np.random.default_rng(9)
        h = (0.03 * rng.uniform(0.6, 1.4, n)).astype(real_dtype(rb))
        tree, counts = oracle.compute_octree(keys, 32)
        o = oracle.build_octree(tree)
        layout = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
        cen, siz = oracle.node_centers(HILBERT, o["prefixes"], box, rb)
        n_ref, c_ref = oracle.find_neighbors(x, y, z, h, 100, n - 50, box, o, layout, cen, siz, ngmax)
        od = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in o.items()}
        n_got, c_got = hip.find_neighbors(dev(x), dev(y), dev(z), dev(h), 100, n - 50, cbox(box), od, dev(layout),
                                          dev(cen), dev(siz), ngmax)
        hip.sync()
        c_got, n_got = host(c_got), host(n_got)
        assert np.array_equal(c_got, c_ref)
        assert c_ref.max() > ngmax // 2  # the case exercises both stored and overflowing lists
        stored = np.minimum(c_ref, ngmax)
        mask = np.arange(ngmax)[None, :] < stored[:, None]
        assert np.array_equal(n_got[mask], n_ref[mask])


@pytest.mark.parametrize("kb", [32, 64])
@pytest.mark.parametrize("curve", [MORTON, HILBERT])
@pytest.mark.parametrize("bc", [(0, 0, 0), (1, 1, 1)])
def test_owner_side_halo_building_blocks(hip, oracle, kb, curve, bc):
    """halo_boxes / find_overlaps (multi-rank halo exchange) against the brute-force checker"""
    box = Box([-1.3, 2.1, 0.2, 0.9, -5, 7], bc)
    x, y, z, keys = _sorted_keys(oracle, curve, kb, 30000, box, 64, seed=51, kind="clustered")
    tree, counts = oracle.compute_octree(keys, 16)
    o = oracle.build_octree(tree)
    od = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in o.items()}
    nl = tree.size - 1
    radii = np.random.default_rng(3).uniform(0.0, 0.08, nl).astype(np.float32)
    for first, last in ((0, nl // 3), (nl // 3, 2 * nl // 3), (2 * nl // 3, nl)):
        b_ref = oracle.halo_boxes(curve, tree, radii, box, first, last)
        b_got = hip.halo_boxes(curve, dev(tree), dev(radii), cbox(box), first, last)
        assert np.array_equal(host(b_got, False), b_ref)
        assert b_ref[:, 6].sum() > 0
        # the boxes exported by one third of the tree are served by the other two thirds
        for f2, l2 in ((0, nl // 3), (nl // 3, 2 * nl // 3), (2 * nl // 3, nl)):
            if f2 == first:
                continue
            f_ref = oracle.find_overlaps(curve, tree, b_ref, f2, l2)
            f_got = hip.find_overlaps(curve, od, dev(tree), b_got, f2, l2)
            hip.sync()
            assert np.array_equal(host(f_got, False), f_ref), (first, f2, f_ref.sum())
            assert f_ref[:f2].sum() == 0 and f_ref[l2:].sum() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kb", [32, 64])
def test_lower_bound_unsigned(hip, kb):
    """lowerBoundGpu range form: unsigned comparison, including the removal marker 2^(3*maxLevel) at the tail"""
    import torch

    rng = np.random.default_rng(5)
    udt = np.uint64 if kb == 64 else np.uint32
    end = 1 << (3 * (21 if kb == 64 else 10))
    keys = np.sort(rng.integers(0, end, 100000, dtype=np.uint64).astype(udt))
    keys = np.concatenate([keys, np.full(7, end, dtype=udt)])
    vals = np.concatenate([rng.integers(0, end, 50, dtype=np.uint64).astype(udt), keys[[0, 5, 99999]],
                           np.array([0, end - 1, end], dtype=udt)])
    sdt = np.int64 if kb == 64 else np.int32
    got = hip.lower_bound(torch.from_numpy(keys.view(sdt)).cuda(), torch.from_numpy(vals.view(sdt)).cuda())
    assert np.array_equal(got.cpu().numpy(), np.searchsorted(keys, vals, side="left"))
    empty = hip.lower_bound(torch.zeros(0, dtype=torch.int64 if kb == 64 else torch.int32, device="cuda"),
                            torch.from_numpy(vals.view(sdt)).cuda())
    assert int(empty.abs().sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kb", [32, 64])
def test_merge_positions_and_gather_scatter(hip, kb):
    """stable merge of two sorted runs (ties: run a first) = np.sort(kind="stable") of their concatenation;
    gather_scatter = dst[map_out] = src[map_in]"""
    import torch

    rng = np.random.default_rng(kb)
    udt = np.uint64 if kb == 64 else np.uint32
    sdt = np.int64 if kb == 64 else np.int32
    end = 1 << (3 * (21 if kb == 64 else 10))
    for na, nb in ((100000, 3000), (5000, 5000), (1, 0), (0, 7), (70000, 1)):
        a = np.sort(rng.integers(0, 5000, na).astype(udt) * udt(end // 5000))  # many ties inside and across the runs
        b = np.sort(rng.integers(0, 5000, nb).astype(udt) * udt(end // 5000))
        pa, pb = hip.merge_positions(torch.from_numpy(a.view(sdt)).cuda(), torch.from_numpy(b.view(sdt)).cuda(), 11)
        pa, pb = pa.cpu().numpy(), pb.cpu().numpy()
        order = np.argsort(np.concatenate([a, b]), kind="stable")
        ref = np.empty(na + nb, dtype=np.int64)
        ref[order] = np.arange(na + nb)
        assert np.array_equal(pa, ref[:na] + 11) and np.array_equal(pb, ref[na:] + 11)
    n = 200001
    for dt in (np.float64, np.float32):
        src = rng.normal(size=n).astype(dt)
        map_in = rng.permutation(n).astype(np.int32)
        map_out = rng.permutation(n).astype(np.int32)
        dst = torch.zeros(n, dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda")
        hip.gather_scatter(torch.from_numpy(map_in).cuda(), torch.from_numpy(map_out).cuda(),
                           torch.from_numpy(src).cuda(), dst)
        ref = np.zeros(n, dtype=dt)
        ref[map_out] = src[map_in]
        assert np.array_equal(dst.cpu().numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("kb", [32, 64])
def test_sort_keys_ordering_equals_sort_pairs_of_iota(hip, kb):
    """setMapFromCodes = sequenceGpu + sortByKeyGpu: same keys and permutation as sorting (key, position) pairs"""
    import torch

    rng = np.random.default_rng(kb + 3)
    for n in (1, 777, 4096, 70001, 3_000_017):
        src = rng.integers(0, 1 << 20, n, dtype=np.int64) << (10 if kb == 64 else 0)
        src = torch.from_numpy(src.astype(np.int64 if kb == 64 else np.int32)).cuda()
        k1, v1 = src.clone(), torch.arange(n, dtype=torch.int32, device="cuda")
        hip.sort_pairs(k1, v1)
        k2 = src.clone()
        v2 = hip.sort_keys_ordering(k2)
        hip.sync()
        assert torch.equal(k1, k2) and torch.equal(v1, v2)


@pytest.mark.gpu
@pytest.mark.parametrize("kb,rb", [(32, 32), (64, 64), (64, 32)])
@pytest.mark.parametrize("curve", [MORTON, HILBERT])
def test_sfc_keys_and_ordering_equals_two_calls(hip, kb, rb, curve):
    """the fused encode + sort (digits counted inside the encode kernel) against compute_sfc_keys + sort_pairs,
    including remove markers, an unaligned start and sizes around the vector width"""
    import torch

    box = Box([-1.0, 2.0, 0.0, 1.0, 3.0, 7.0], (0, 1, 0))
    for n, off in ((1, 0), (2, 0), (1001, 0), (250003, 0), (250003, 1), (2_000_001, 0)):
        x, y, z = random_cloud(n + off, box, rb, seed=n, kind="clustered")
        xd, yd, zd = [dev(a)[off:] for a in (x, y, z)]
        marked = torch.zeros(n, dtype=torch.int64 if kb == 64 else torch.int32, device="cuda")
        marked[::97] = -(1 << 63) if kb == 64 else 1 << 30  # remove markers (bit pattern 2^(3 maxLevel)) must survive
        k1 = hip.compute_sfc_keys(curve, kb, xd, yd, zd, cbox(box), keys=marked.clone())
        v1 = torch.arange(n, dtype=torch.int32, device="cuda")
        hip.sort_pairs(k1, v1)
        k2, v2 = hip.sfc_keys_and_ordering(curve, kb, xd, yd, zd, cbox(box), keys=marked.clone())
        hip.sync()
        assert torch.equal(k1, k2) and torch.equal(v1, v2)


@pytest.mark.gpu
@pytest.mark.parametrize("kb", [32, 64])
def test_node_counts_guided_any_guess(hip, oracle, kb):
    """useCountsAsGuess: the result never depends on the guess (exact, shifted, zero, beyond the end, random)"""
    import torch

    box = Box([0, 1, 0, 1, 0, 1])
    x, y, z, keys = _sorted_keys(oracle, HILBERT, kb, 300000, box, 64, seed=77, kind="clustered")
    tree, counts = oracle.compute_octree(keys, 32)
    nl = counts.size
    exact = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    rng = np.random.default_rng(1)
    kd, td = dev(keys), dev(tree)
    for guess in (exact, np.clip(exact.astype(np.int64) + rng.integers(-40, 40, nl + 1), 0, keys.size).astype(np.uint32),
                  np.zeros(nl + 1, np.uint32), np.full(nl + 1, 0xFFFFFFFF, np.uint32),
                  rng.integers(0, keys.size, nl + 1).astype(np.uint32)):
        got = hip.compute_node_counts_guided(td, kd, dev(guess))
        assert np.array_equal(host(got), counts)
    # keys confined to the middle of the tree: leaves left and right of the populated range count zero
    sub = keys[keys.size // 3: keys.size // 2]
    ref = oracle.node_counts(tree, sub)
    got = hip.compute_node_counts_guided(td, dev(sub), dev(exact))
    assert np.array_equal(host(got), ref)


@pytest.mark.gpu
def test_gather_vec3_double_at_8_byte_alignment(hip):
    """Vec3<double> (24-byte elements, alignas 8 in the reference, util/array.hpp:42-58): arrays that are only 8-byte
    aligned must be accepted"""
    import torch

    n = 50000
    rng = np.random.default_rng(8)
    src = torch.from_numpy(rng.normal(size=(n + 1, 3))).cuda()
    perm = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
    s = src[1:]                      # base + 24 bytes: 8-byte aligned, not 16
    assert s.data_ptr() % 16 == 8
    dst = torch.zeros(n + 1, 3, dtype=torch.float64, device="cuda")[1:]
    hip.gather(perm, s, dst, elem_bytes=24)
    hip.sync()
    assert torch.equal(dst, s[perm.long()])


@pytest.mark.gpu
@pytest.mark.parametrize("kb", [32, 64])
def test_sort_adversarial_digit_patterns_large_tiles(hip, kb):
    """the 16 Ki-pair tile kernel (n >= 2^20) on inputs that stress the rank by LDS atomics: two alternating values, runs of
    32 and 64 equal digits, a single value, few values in every byte, descending order; checked against torch's stable sort"""
    import torch

    n = 1_100_003  # 67 full tiles + a tail
    dt = torch.int64 if kb == 64 else torch.int32
    i = torch.arange(n, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(kb)
    top = (1 << 62) if kb == 64 else (1 << 30)
    nbytes = 8 if kb == 64 else 4
    rep = lambda v: sum(int(v) << (8 * b) for b in range(nbytes - 1))  # the same byte in every digit position  # noqa: E731
    patterns = {
        "alternating": torch.where(i % 2 == 0, rep(0x11), rep(0xEE)),
        "runs32": ((i // 32) % 5) * rep(0x21),
        "runs64": ((i // 64) % 3) * rep(0x7F) + (i % 2),
        "constant": torch.full((n,), rep(0x5A)),
        "few_values": torch.randint(0, 4, (n,), device="cuda", generator=g) * rep(0x33),
        "descending": top - 1 - i * ((top - 1) // n),
        "lane_pattern": ((i % 64) // 16) * rep(0x40) + ((i // 64) % 2),
    }
    for name, k in patterns.items():
        keys = k.to(device="cuda", dtype=dt).contiguous()
        ref_k, ref_v = torch.sort(keys, stable=True)
        vals = torch.arange(n, dtype=torch.int32, device="cuda")
        hip.sort_pairs(keys, vals)
        hip.sync()
        assert torch.equal(keys, ref_k), name
        assert torch.equal(vals.long(), ref_v), name


@pytest.mark.gpu
@pytest.mark.parametrize("kb", [32, 64])
def test_sort_pairs_fuzz_sizes(hip, kb):
    """150 seeded random sizes between 1 and 3.3e6 (both tile shapes, partial tiles, tails of a few elements), keys with
    a random number of significant bits: stable-sorted keys and permutation equal torch's stable sort"""
    import torch

    rng = np.random.default_rng(100 + kb)
    dt = torch.int64 if kb == 64 else torch.int32
    maxbits = 62 if kb == 64 else 30
    g = torch.Generator(device="cuda").manual_seed(kb)
    sizes = [int(v) for v in rng.integers(1, 3_300_000, 120)] + [16384 * k + d for k in (1, 64, 65) for d in (-1, 0, 1)] + \
            [4096 * k + d for k in (1, 255, 256) for d in (-1, 0, 1)] + [1, 2, 63, 64, 65, 1 << 20, (1 << 20) - 1, (1 << 20) + 1]
    for n in sizes:
        bits = int(rng.integers(1, maxbits + 1))
        keys = torch.randint(0, 1 << bits, (n,), dtype=torch.int64, device="cuda", generator=g).to(dt)
        ref_k, ref_v = torch.sort(keys, stable=True)
        vals = torch.arange(n, dtype=torch.int32, device="cuda")
        hip.sort_pairs(keys, vals)
        hip.sync()
        assert torch.equal(keys, ref_k), (n, bits)
        assert torch.equal(vals.long(), ref_v), (n, bits)


@pytest.mark.gpu
@pytest.mark.parametrize("kb", [32, 64])
@pytest.mark.parametrize("curve", [MORTON, HILBERT])
@pytest.mark.parametrize("bc", [(0, 0, 0), (1, 1, 1)])
def test_halo_boxes_foreign_flags_only_boxes_that_leave_the_range(hip, oracle, kb, curve, bc):
    """the exporter's filter: same boxes as halo_boxes, record[6] set exactly for those that overlap a leaf outside the
    own leaf range (brute-force overlap checker of the oracle, one box at a time), never for a box halo_boxes lets go"""
    box = Box([-1.3, 2.1, 0.2, 0.9, -5, 7], bc)
    x, y, z, keys = _sorted_keys(oracle, curve, kb, 20000, box, 64, seed=61, kind="clustered")
    tree, counts = oracle.compute_octree(keys, 16)
    o = oracle.build_octree(tree)
    od = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in o.items()}
    nl = tree.size - 1
    radii = np.random.default_rng(4).uniform(0.0, 0.05, nl).astype(np.float32)
    for first, last in ((0, nl // 3), (nl // 3, 2 * nl // 3), (2 * nl // 3, nl)):
        plain = host(hip.halo_boxes(curve, dev(tree), dev(radii), cbox(box), first, last), False)
        got = host(hip.halo_boxes_foreign(curve, od, dev(tree), dev(radii), cbox(box), first, last), False)
        assert np.array_equal(got[:, :6], plain[:, :6]) and np.all(got[:, 7] == 0)
        assert np.all(got[:, 6] <= plain[:, 6]) and 0 < got[:, 6].sum() < plain[:, 6].sum()
        rng = np.random.default_rng(first)
        for k in rng.choice(last - first, 150, replace=False):
            rec = plain[k:k + 1].copy()
            rec[0, 6] = 1
            outside = 0
            if first > 0:
                outside += int(oracle.find_overlaps(curve, tree, rec, 0, first).sum())
            if last < nl:
                outside += int(oracle.find_overlaps(curve, tree, rec, last, nl).sum())
            assert int(got[k, 6]) == (1 if outside else 0), (first, int(k), outside)


@pytest.mark.gpu
@pytest.mark.parametrize("kb", [32, 64])
def test_sort_with_ballot_ranking(hip, kb, monkeypatch):
    """the ranking variant that does not rely on the service order of returning LDS atomics (what a device that failed
    the one-time probe would run; forced here with CSTONE_SORT_BALLOT_RANK): the fuzzed sizes and the adversarial digit
    patterns give torch's stable sort"""
    monkeypatch.setenv("CSTONE_SORT_BALLOT_RANK", "1")
    test_sort_adversarial_digit_patterns_large_tiles(hip, kb)
    test_sort_pairs_fuzz_sizes(hip, kb)


@pytest.mark.parametrize("elem_dtype,num", [("float32", 2), ("float64", 3), ("float64", 4), ("complex128", 3)])
def test_gather_multi(hip, elem_dtype, num):
    """cstone_hip_gather_multi: several arrays through one map (gatherArrays, R/domain/layout.hpp:203-239, array after
    array in the reference) against torch indexing, sizes around the kernel's 1024-element blocks"""
    import ctypes as C

    import torch

    dt = getattr(torch, elem_dtype)
    for n in (1, 255, 1024, 1025, 70001):
        g = torch.Generator(device="cuda").manual_seed(n)
        perm = torch.randperm(n, device="cuda", generator=g).to(torch.int32)
        src = [torch.randn(n, device="cuda", generator=g, dtype=torch.float64).to(dt) + a for a in range(num)]
        dst = [torch.zeros_like(t) for t in src]
        sp = (C.c_void_p * num)(*[t.data_ptr() for t in src])
        dp = (C.c_void_p * num)(*[t.data_ptr() for t in dst])
        rc = hip.lib.cstone_hip_gather_multi(hip.h, C.c_int(src[0].element_size()), C.c_void_p(perm.data_ptr()),
                                             C.c_size_t(n), sp, dp, C.c_int(num))
        hip._chk(rc, "gather_multi")
        hip.sync()
        for a in range(num):
            assert torch.equal(dst[a], src[a][perm.long()]), (n, a)
    # a destination that is also a source is refused
    sp = (C.c_void_p * 2)(src[0].data_ptr(), src[1].data_ptr())
    dp = (C.c_void_p * 2)(src[1].data_ptr(), dst[0].data_ptr())
    assert hip.lib.cstone_hip_gather_multi(hip.h, C.c_int(src[0].element_size()), C.c_void_p(perm.data_ptr()),
                                           C.c_size_t(n), sp, dp, C.c_int(2)) != 0


def test_find_neighbors_stats(hip):
    """cstone_hip_find_neighbors_stats: the counters of the reference's NcStats (find_neighbors.cuh:345-369).  On a tree
    that is one leaf every target is tested against every particle: sumP2P = targets * n exactly, nothing on the stack; on
    a real tree the counters bracket the neighbour counts"""
    import ctypes as C

    import torch

    import cstone_amd

    n = 20000
    g = torch.Generator(device="cuda").manual_seed(5)
    x, y, z = [torch.rand(n, dtype=torch.float64, device="cuda", generator=g) for _ in range(3)]
    h = torch.full((n,), 0.02, dtype=torch.float64, device="cuda")
    cb = cstone_amd.make_cbox([0, 1] * 3)
    keys = hip.compute_sfc_keys(cstone_amd.HILBERT, 64, x, y, z, cb)
    order = torch.arange(n, dtype=torch.int32, device="cuda")
    hip.sort_pairs(keys, order)
    x, y, z = [a[order.long()].contiguous() for a in (x, y, z)]
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def run(bucket, first, last):
        tree, counts, _ = hip.compute_octree(keys, bucket)
        o = hip.build_octree(tree)
        layout = torch.cat([torch.zeros(1, dtype=torch.int32, device="cuda"), counts.cumsum(0).to(torch.int32)])
        cen, siz = hip.node_centers(cstone_amd.HILBERT, o["prefixes"], cb, 64)
        nc = torch.zeros(last - first, dtype=torch.int32, device="cuda")
        st = (C.c_uint64 * 4)()
        rc = hip.lib.cstone_hip_find_neighbors_stats(hip.h, C.c_int(64), P(x), P(y), P(z), P(h), C.c_uint32(first),
                                                     C.c_uint32(last), C.byref(cb), P(o["child_offsets"]),
                                                     P(o["internal_to_leaf"]), P(layout), P(cen), P(siz), C.c_float(1.0),
                                                     C.c_uint32(0), None, P(nc), st)
        hip._chk(rc, "find_neighbors_stats")
        _, nc_plain = hip.find_neighbors(x, y, z, h, first, last, cb, o, layout, cen, siz, 0)
        hip.sync()
        assert torch.equal(nc, nc_plain)  # the instrumented kernel finds the same neighbours
        return [int(v) for v in st], nc

    st, nc = run(n + 1, 100, 1124)  # the root is the only leaf
    assert st[0] == 1024 * n and st[1] == n and st[2] == 0 and st[3] == 1024 * n
    st, nc = run(64, 0, n)
    total = int(nc.long().sum().item())
    assert total + n <= st[0] <= st[3] and st[1] <= st[0] and 0 < st[2] <= 160
    assert st[1] >= int(nc.max().item()) + 1


@pytest.mark.gpu
def test_wide_and_narrow_flavours_of_the_primitives(hip):
    """the instantiations of the reference's list that the shim used to stage through the host
    (R/primitives/primitives_gpu.cu: lowerBoundGpu with 32-bit results :214-238, sequenceGpu<uint64_t> :88-103,
    exclusive / inclusiveScanGpu of 32-bit values with 64-bit sums :395-437) against numpy"""
    import ctypes as C

    import torch

    rng = np.random.default_rng(11)
    for kb, kdt, tdt in ((32, np.uint32, torch.int32), (64, np.uint64, torch.int64)):
        keys = np.sort(rng.integers(0, 1 << (kb - 2), 50_001).astype(kdt))
        vals = rng.integers(0, 1 << (kb - 2), 3_333).astype(kdt)
        vals[:3] = (0, keys[-1], keys[-1] + 1)
        dk = torch.from_numpy(keys.view(np.int32 if kb == 32 else np.int64)).cuda()
        dv = torch.from_numpy(vals.view(np.int32 if kb == 32 else np.int64)).cuda()
        out = torch.zeros(vals.size, dtype=torch.int32, device="cuda")
        hip._chk(hip.lib.cstone_hip_lower_bound_u32(hip.h, C.c_int(kb), C.c_void_p(dk.data_ptr()), C.c_size_t(keys.size),
                                                    C.c_void_p(dv.data_ptr()), C.c_int(vals.size),
                                                    C.c_void_p(out.data_ptr())), "lower_bound_u32")
        hip.sync()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), np.searchsorted(keys, vals, side="left").astype(np.uint32))
    for n in (1, 255, 2048, 2049, 1_000_003):
        seq = torch.zeros(n, dtype=torch.int64, device="cuda")
        init = (1 << 40) + 5
        hip._chk(hip.lib.cstone_hip_sequence_u64(hip.h, C.c_void_p(seq.data_ptr()), C.c_size_t(n), C.c_uint64(init)), "seq")
        vals = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)  # sums beyond 32 bits from a few elements on
        din = torch.from_numpy(vals.view(np.int32)).cuda()
        for inclusive in (0, 1):
            dout = torch.zeros(n, dtype=torch.int64, device="cuda")
            hip._chk(hip.lib.cstone_hip_scan_u32_to_u64(hip.h, C.c_void_p(din.data_ptr()), C.c_void_p(dout.data_ptr()),
                                                        C.c_size_t(n), C.c_uint64(7), C.c_int(inclusive)), "scan64")
            hip.sync()
            c = np.cumsum(vals.astype(np.uint64)) + np.uint64(7)
            want = c if inclusive else np.concatenate(([np.uint64(7)], c[:-1]))
            assert np.array_equal(dout.cpu().numpy().view(np.uint64), want), (n, inclusive)
        hip.sync()
        assert np.array_equal(seq.cpu().numpy().view(np.uint64), np.arange(n, dtype=np.uint64) + np.uint64(init))


@pytest.mark.gpu
def test_find_neighbors_interleaved_lists(hip):
    """cstone_hip_find_neighbors_interleaved: the same neighbours in the layout of the reference's warp-interleaved lists
    (traverseNeighbors, R/traversal/find_neighbors.cuh:116 with targetSize = 64): entry ((t / 64) * ngmax + k) * 64 + t % 64
    equals entry t * ngmax + k of the row-major lists, for a range that neither starts nor ends on a multiple of 64"""
    import ctypes as C

    import torch

    import cstone_amd

    n, ngmax = 30000, 48
    g = torch.Generator(device="cuda").manual_seed(9)
    x, y, z = [torch.rand(n, dtype=torch.float64, device="cuda", generator=g) for _ in range(3)]
    h = torch.full((n,), 0.02, dtype=torch.float64, device="cuda")
    cb = cstone_amd.make_cbox([0, 1] * 3, [1, 0, 1])
    keys = hip.compute_sfc_keys(cstone_amd.HILBERT, 64, x, y, z, cb)
    order = torch.arange(n, dtype=torch.int32, device="cuda")
    hip.sort_pairs(keys, order)
    x, y, z = [a[order.long()].contiguous() for a in (x, y, z)]
    tree, counts, _ = hip.compute_octree(keys, 64)
    o = hip.build_octree(tree)
    layout = torch.cat([torch.zeros(1, dtype=torch.int32, device="cuda"), counts.cumsum(0).to(torch.int32)])
    cen, siz = hip.node_centers(cstone_amd.HILBERT, o["prefixes"], cb, 64)
    first, last = 37, n - 11
    nt = last - first
    rows, nc = hip.find_neighbors(x, y, z, h, first, last, cb, o, layout, cen, siz, ngmax)
    blocks = (nt + 63) // 64
    inter = torch.full((blocks * 64 * ngmax,), -1, dtype=torch.int32, device="cuda")
    nc2 = torch.zeros(nt, dtype=torch.int32, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    rc = hip.lib.cstone_hip_find_neighbors_interleaved(hip.h, C.c_int(64), P(x), P(y), P(z), P(h), C.c_uint32(first),
                                                       C.c_uint32(last), C.byref(cb), P(o["child_offsets"]),
                                                       P(o["internal_to_leaf"]), P(layout), P(cen), P(siz), C.c_float(1.0),
                                                       C.c_uint32(ngmax), P(inter), P(nc2))
    hip._chk(rc, "find_neighbors_interleaved")
    hip.sync()
    assert torch.equal(nc, nc2)
    rows = rows.reshape(nt, ngmax).cpu().numpy()
    got = inter.reshape(blocks, ngmax, 64).permute(0, 2, 1).reshape(blocks * 64, ngmax)[:nt].cpu().numpy()
    cnt = np.minimum(nc.cpu().numpy(), ngmax)
    valid = np.arange(ngmax)[None, :] < cnt[:, None]
    assert valid.sum() > 5 * nt and np.array_equal(rows[valid], got[valid])
    assert (got[~valid] == -1).all()  # nothing is written beyond a target's count
