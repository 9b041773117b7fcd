"""Domain::sync parity (GPU): cstone_hip_domain_* against fixtures produced by the reference's own
cstone::Domain<KeyType,T,CpuTag> (all four combinations of 32-/64-bit keys and float/double) on one rank (tests/golden/make_golden_domain.py), step by step with moving
particles, shrinking boxes, periodic axes and particle removal.  Everything is compared bit-for-bit, exactly as the
reference's own GPU-vs-CPU integration test does (test/integration_mpi/domain_gpu.cpp:117-136)."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_domain_*.npz"))
              if "_mpi_" not in os.path.basename(p))  # the multi-rank fixtures belong to test_distributed.py


@pytest.mark.gpu
@pytest.mark.parametrize("full_sort,num_scratch", [(False, 1), (True, 1), (False, 3), (False, 4)],
                         ids=["partial-digit-sort", "all-digits-sorted", "three-scratch-buffers", "four-scratch-buffers"])
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_domain_sync_matches_reference(hip, path, full_sort, num_scratch, monkeypatch):
    import torch

    import cstone_amd
    from cstone_amd.domain import Domain

    d = np.load(path)
    kb = int(d["key_bits"]) if "key_bits" in d else 64
    rb = int(d["real_bits"]) if "real_bits" in d else 64
    kdt, ksigned = (np.uint64, np.int64) if kb == 64 else (np.uint32, np.int32)
    box = cstone_amd.make_cbox(d["lim"], d["bc"])
    dom = Domain(hip, cstone_amd.HILBERT, kb, rb, int(d["bucket"]), int(d["bucket_focus"]), 0.5, box)
    if full_sort:  # the radix passes over ALL key digits instead of the digits above the previous tree's leaf level
        dom.set_sort_mode(dom.SORT_ALL_DIGITS)
    for s in range(int(d["steps"])):
        x, y, z, h = [torch.from_numpy(d[f"in{s}_{c}"].copy()).cuda() for c in "xyzh"]
        assert x.element_size() * 8 == rb
        n = x.numel()
        kin = d[f"in{s}_keys"] if f"in{s}_keys" in d else np.zeros(n, kdt)
        keys = torch.from_numpy(kin.view(ksigned).copy()).cuda()
        # one scratch buffer, or the scratch tuple of the reference's sync (cstone_hip_domain_sync_scratch: x, y, z are then
        # gathered by one kernel)
        scratch = torch.empty_like(x) if num_scratch == 1 else [torch.empty_like(x) for _ in range(num_scratch)]
        # a conserved property travelling along (not wider than the coordinates: it shares their scratch buffer)
        tag = torch.arange(n, dtype=torch.float64 if rb == 64 else torch.float32, device="cuda")
        late = [torch.stack([x, y, z], dim=1).to(torch.float32).contiguous(), (h * 1e3).to(torch.int16)]
        keys, x, y, z, h, scratch, props = dom.sync(keys, x, y, z, h, scratch, [tag])
        # reapplySync (domain.hpp:334-378): fields that were not part of the sync are brought into the new order later
        for f, w in zip(late, [torch.stack([x, y, z], dim=1).to(torch.float32), (h * 1e3).to(torch.int16)]):
            assert torch.equal(dom.reapply_sync(f), w), s
        with pytest.raises(cstone_amd.CstoneError, match="last sync took"):
            dom.reapply_sync(late[1][:-1])
        hip.sync()
        v = dom.view()
        info = d[f"out{s}_info"]
        assert [v.start_index, v.end_index, v.num_particles_with_halos, v.num_global_leaves, v.num_focus_leaves] == \
            info.tolist(), s
        assert np.array_equal(np.array(list(v.box.lim)), d[f"out{s}_box"]), s
        m = int(info[2])
        assert np.array_equal(keys.cpu().numpy().view(kdt), d[f"out{s}_keys"])
        for t, c in ((x, "x"), (y, "y"), (z, "z"), (h, "h")):
            assert np.array_equal(t.cpu().numpy(), d[f"out{s}_{c}"]), (s, c)
        # the property followed its particle: tag[i] is the input index of output particle i
        src = props[0].cpu().numpy().astype(np.int64)
        assert np.array_equal(d[f"in{s}_x"][src], d[f"out{s}_x"])
        ngl, nfl = int(info[3]), int(info[4])
        assert np.array_equal(dom.fetch(v.global_leaves, ngl + 1, kdt), d[f"out{s}_global_leaves"])
        assert np.array_equal(dom.fetch(v.focus_leaves, nfl + 1, kdt), d[f"out{s}_focus_leaves"])
        assert np.array_equal(dom.fetch(v.focus_leaf_counts, nfl, np.uint32), d[f"out{s}_focus_counts"])
        assert np.array_equal(dom.fetch(v.layout, nfl + 1, np.uint32), d[f"out{s}_layout"])
        assert dom.fetch(v.halo_flags, nfl, np.int32).sum() == 0  # one rank: no halos
        assert m == int(dom.fetch(v.layout, nfl + 1, np.uint32)[-1])


@pytest.mark.gpu
def test_domain_argument_errors(hip):
    import cstone_amd
    from cstone_amd.domain import Domain

    with pytest.raises(cstone_amd.CstoneError, match="bucket size of the global tree"):
        Domain(hip, cstone_amd.HILBERT, 64, 64, 8, 64)  # domain.hpp:108-112
    with pytest.raises(cstone_amd.CstoneError, match="single-rank"):
        Domain(hip, cstone_amd.HILBERT, 64, 64, 64, 8, rank=0, nranks=2)


@pytest.mark.gpu
def test_sync_reports_the_device_side_error_word(hip, monkeypatch):
    """the sticky device-side check word (look-back spin bail-out of the sort, traversal stack overflow) must reach the
    caller of sync: with the word forced, cstone_hip_domain_sync returns CSTONE_E_INTERNAL once, and the re-armed word
    lets the next sync pass"""
    import torch
    import cstone_amd
    from cstone_amd.domain import Domain

    n = 5000
    rng = np.random.default_rng(3)
    cb = cstone_amd.make_cbox([0, 1, 0, 1, 0, 1], [0, 0, 0])
    dom = Domain(hip, cstone_amd.HILBERT, 64, 64, 64, 16, 0.5, cb)
    x, y, z = [torch.from_numpy(rng.uniform(0, 1, n)).cuda() for _ in range(3)]
    h = torch.full((n,), 0.01, dtype=torch.float64, device="cuda")
    keys = torch.zeros(n, dtype=torch.int64, device="cuda")
    # the product library carries no fault injection: the hook lives in the tests' build of the same sources
    monkeypatch.setenv("CSTONE_HIP_LIB", cstone_amd.LIBPATH.replace("libcstone_hip.so", "libcstone_hip_hooks.so"))
    hooked = cstone_amd.Context(0)
    monkeypatch.delenv("CSTONE_HIP_LIB")
    assert hooked.lib.cstone_hip_test_hooks() == 1 and hip.lib.cstone_hip_test_hooks() == 0
    dom_h = Domain(hooked, cstone_amd.HILBERT, 64, 64, 64, 16, 0.5, cb)
    monkeypatch.setenv("CSTONE_FORCE_DEVICE_ERROR", "1")
    with pytest.raises(cstone_amd.CstoneError, match="device-side check failed"):
        dom_h.sync(keys, x.clone(), y.clone(), z.clone(), h.clone(), torch.empty_like(x))
    # (the product build ignores the variable)
    dom.sync(keys.clone(), x.clone(), y.clone(), z.clone(), h.clone(), torch.empty_like(x))
    hip.sync()
    monkeypatch.delenv("CSTONE_FORCE_DEVICE_ERROR")
    dom2 = Domain(hip, cstone_amd.HILBERT, 64, 64, 64, 16, 0.5, cb)
    out = dom2.sync(keys, x.clone(), y.clone(), z.clone(), h.clone(), torch.empty_like(x))
    hip.sync()
    assert out[0].numel() == n


@pytest.mark.gpu
def test_partial_sort_fallback_when_particles_collapse(hip, oracle):
    """Domain::sync radix-sorts only the key digits above the previous tree's depth and orders the rest inside runs of
    equal high digits; when the particles suddenly collapse into a tiny region those runs become too long and the
    regular sort must take over.  Keys, ordering and attached fields must equal the oracle's full stable sort."""
    import torch

    import cstone_amd
    from cstone_amd.domain import Domain
    from oracle.oracle import HILBERT, Box

    n = 400000
    rng = np.random.default_rng(3)
    x, y, z = [rng.uniform(0, 1, n) for _ in range(3)]
    h = np.full(n, 0.005)
    dom = Domain(hip, cstone_amd.HILBERT, 64, 64, 4096, 64, 0.5, cstone_amd.make_cbox([0, 1] * 3, (1, 1, 1)))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    xd, yd, zd, hd = dev(x), dev(y), dev(z), dev(h)
    keys = torch.zeros(n, dtype=torch.int64, device="cuda")
    scratch = torch.empty(n, dtype=torch.float64, device="cuda")
    ident = torch.arange(n, dtype=torch.float64, device="cuda")
    box = Box([0, 1] * 3, (1, 1, 1))
    for step in range(4):
        if step == 2:
            # collapse: everything within 1e-7 of one point, thousands of particles share 40+ key bits
            xd = 0.3 + (xd - 0.5) * 2e-7
            yd = 0.6 + (yd - 0.5) * 2e-7
            zd = 0.2 + (zd - 0.5) * 2e-7
        xin, yin, zin, idin = [t.cpu().numpy().copy() for t in (xd, yd, zd, ident)]
        keys, xd, yd, zd, hd, scratch, (ident,) = dom.sync(keys, xd, yd, zd, hd, scratch, [ident])
        hip.sync()
        kref = oracle.compute_sfc_keys(HILBERT, 64, xin, yin, zin, box)
        ks, order = oracle.sort_pairs(kref, np.arange(n))
        assert np.array_equal(keys.cpu().numpy().view(np.uint64), ks), step
        assert np.array_equal(ident.cpu().numpy(), idin[order]), step   # the same stable permutation
        assert np.array_equal(xd.cpu().numpy(), xin[order]), step


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(10)) + [100, 101])
def test_domain_sync_random_configurations_against_oracle(hip, oracle, seed):
    """Domain::sync over seeded random configurations the reference fixtures do not reach (Morton keys, tiny and huge
    buckets, anisotropic boxes, all boundary combinations, 32/64-bit keys, float/double): after every sync the keys are
    the sorted oracle keys under the domain's box, every field follows its particle, and both trees advance exactly like
    the oracle's update rule (converged from the root on the first call, one rebalance step per later call)"""
    import torch

    import cstone_amd
    from cstone_amd.domain import Domain
    from oracle.oracle import HILBERT, MORTON, Box

    rng = np.random.default_rng(500 + seed)
    kb, rb = int(rng.choice([32, 64])), int(rng.choice([32, 64]))
    curve = int(rng.choice([HILBERT, MORTON]))
    n = int(rng.choice([300, 5000, 60000, 200000]))
    if seed >= 100:
        # 3e6 particles: the 16 Ki-pair tiles of the radix sort, the partial digit passes and the run fix-up inside
        # Domain::sync; seed 100 with 30-bit keys of a clustered cloud (many equal keys: the order among them must be
        # the stable one), seed 101 with 63-bit keys
        n, kb = 3_000_000, (32 if seed == 100 else 64)
    bucket_focus = int(rng.choice([1, 8, 64, 1000]))
    if seed >= 100:
        bucket_focus = 64
    bucket = int(bucket_focus * rng.choice([1, 4, 50]))
    bc = tuple(int(v) for v in rng.integers(0, 3, 3))  # open, periodic, fixed
    lo = rng.uniform(-3, 0, 3)
    hi = lo + rng.uniform(0.5, 4, 3)
    lim = [lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]]
    rdt, kdt, ksigned = (np.float64 if rb == 64 else np.float32), (np.uint64 if kb == 64 else np.uint32), \
        (np.int64 if kb == 64 else np.int32)
    if rng.uniform() < 0.5 and seed != 100:
        pos = rng.uniform(lo, hi, (n, 3))
    else:
        centers = rng.uniform(lo, hi, (4, 3))
        pos = np.clip(centers[rng.integers(0, 4, n)] + rng.normal(0, (hi - lo) / 40, (n, 3)), lo, hi)
    x, y, z = [np.ascontiguousarray(pos[:, d]).astype(rdt) for d in range(3)]
    h = rng.uniform(0.001, 0.01, n).astype(rdt)
    dom = Domain(hip, curve, kb, rb, bucket, bucket_focus, 0.5, cstone_amd.make_cbox(lim, bc))
    ftree = gtree = None
    for s in range(3):
        xd, yd, zd, hd = [torch.from_numpy(a.copy()).cuda() for a in (x, y, z, h)]
        m = x.size
        keys = torch.zeros(m, dtype=torch.int64 if kb == 64 else torch.int32, device="cuda")
        scratch = torch.empty_like(xd)
        ident = torch.arange(m, dtype=torch.float64 if rb == 64 else torch.float32, device="cuda")
        keys, xd, yd, zd, hd, scratch, (ident,) = dom.sync(keys, xd, yd, zd, hd, scratch, [ident])
        hip.sync()
        v = dom.view()
        assert (v.start_index, v.end_index, v.num_particles_with_halos) == (0, m, m)
        box = Box(list(v.box.lim), bc)
        want = oracle.compute_sfc_keys(curve, kb, x, y, z, box)
        order = np.argsort(want, kind="stable")
        got_keys = keys.cpu().numpy().view(kdt)
        assert np.array_equal(got_keys, want[order]), s
        src = ident.cpu().numpy().astype(np.int64)
        assert np.array_equal(src, order), s  # the stable permutation itself
        for a, ad in ((x, xd), (y, yd), (z, zd), (h, hd)):
            assert np.array_equal(ad.cpu().numpy(), a[order]), s
        # global tree: converged from the root on the first call, one rebalance step (csarray.hpp:430-448) per later call
        ks = want[order]
        if s == 0:
            gtree, gcounts = oracle.compute_octree(ks, bucket)
        else:
            gtree, gcounts, _ = oracle.update_octree(ks, bucket, gtree, gcounts)
        assert np.array_equal(dom.fetch(v.global_leaves, v.num_global_leaves + 1, kdt), gtree), s
        # focus tree: converged on the first call; later calls follow the focus rule (one level per step, merges decided
        # on the parents, focus/rebalance.hpp:50-184), which the reference fixtures pin -- here: a valid cornerstone leaf
        # array whose counts are those of the keys
        got_f = dom.fetch(v.focus_leaves, v.num_focus_leaves + 1, kdt)
        got_c = dom.fetch(v.focus_leaf_counts, v.num_focus_leaves, np.uint32)
        if s == 0:
            ftree, fcounts = oracle.compute_octree(ks, bucket_focus)
            assert np.array_equal(got_f, ftree) and np.array_equal(got_c, fcounts)
        assert got_f[0] == 0 and int(got_f[-1]) == 1 << (3 * (21 if kb == 64 else 10)) and np.all(got_f[1:] > got_f[:-1])
        span = np.diff(got_f.astype(np.uint64) if kb == 64 else got_f.astype(np.uint64))
        lvl = np.log2(span.astype(np.float64)) / 3
        assert np.all(lvl == np.round(lvl)) and np.all(got_f[:-1].astype(np.uint64) % span == 0)  # octree nodes
        assert np.array_equal(got_c, oracle.node_counts(got_f, ks)), s
        assert np.array_equal(dom.fetch(v.layout, v.num_focus_leaves + 1, np.uint32),
                              np.concatenate([[0], np.cumsum(got_c)]).astype(np.uint32))
        # move a little (inside the box for fixed / periodic axes)
        x, y, z, h = [a[order] for a in (x, y, z, h)]
        for a, d in ((x, 0), (y, 1), (z, 2)):
            a += (rng.normal(0, 0.003, m) * (hi[d] - lo[d])).astype(rdt)
            np.clip(a, rdt(lo[d]), np.nextafter(rdt(hi[d]), rdt(lo[d])), out=a)


@pytest.mark.gpu
@pytest.mark.parametrize("kb,rb,mass_bits", [(64, 64, 64), (64, 64, 32), (32, 32, 32)])
def test_domain_sync_grav_expansion_centers_equal_the_oracle(hip, oracle, kb, rb, mass_bits):
    """Domain::syncGrav on one rank (cstone_hip_domain_sync_grav): the tree is the tree of sync, the masses follow their
    particles, and (centre of mass, MAC radius^2) of every node equal computeLeafSourceCenter + the CombineSourceCenter
    upsweep + setMac of the oracle (pinned against the reference's own functions in tests/test_focus.py) bit for bit --
    over three syncs with moving particles; updateExpansionCenters afterwards with other masses likewise"""
    import torch

    import cstone_amd
    from cstone_amd.domain import Domain
    from oracle.oracle import HILBERT, Box, max_level

    rng = np.random.default_rng(77 + kb + rb + mass_bits)
    n, bucket_focus, theta = 40000, 16, 0.58
    rdt = np.float64 if rb == 64 else np.float32
    mdt = np.float64 if mass_bits == 64 else np.float32
    kdt = np.uint64 if kb == 64 else np.uint32
    bc = (0, 1, 0)
    lim = [-1.0, 1.0, 0.0, 2.0, -0.5, 0.5]
    centers = rng.uniform(-0.5, 0.5, (5, 3))
    pos = centers[rng.integers(0, 5, n)] + rng.normal(0, 0.08, (n, 3))
    x = np.clip(pos[:, 0], -0.99, 0.99).astype(rdt)
    y = np.clip(pos[:, 1] + 1.0, 0.01, 1.99).astype(rdt)
    z = np.clip(pos[:, 2], -0.49, 0.49).astype(rdt)
    h = rng.uniform(0.002, 0.01, n).astype(rdt)
    m = rng.uniform(0.5, 2.0, n).astype(mdt)
    dom = Domain(hip, HILBERT, kb, rb, 64 * bucket_focus, bucket_focus, theta, cstone_amd.make_cbox(lim, bc))
    same_tree = Domain(hip, HILBERT, kb, rb, 64 * bucket_focus, bucket_focus, theta, cstone_amd.make_cbox(lim, bc))

    def expect(v, xs, ys, zs, ms):
        L, M = v.num_focus_leaves, v.num_focus_nodes
        octree = dict(prefixes=dom.fetch(v.prefixes, M, kdt), child_offsets=dom.fetch(v.child_offsets, M + 1, np.int32),
                      level_range=dom.fetch(v.level_range, max_level(kb) + 2, np.int32))
        l2i = dom.fetch(v.leaf_to_internal, M, np.int32)[M - L:]
        layout = dom.fetch(v.layout, L + 1, np.uint32)
        ctr = oracle.leaf_source_centers(xs, ys, zs, ms, l2i, layout, M, rb)
        ctr = oracle.upsweep_centers(octree, ctr, max_level(kb))
        box = Box(list(v.box.lim), bc)
        # (1 / theta in float arithmetic, like the library and the reference: 1.0f / theta_)
        return oracle.mac_spheres(HILBERT, 1, octree["prefixes"], box, float(np.float32(1.0) / np.float32(theta)), rb, ctr)

    for s in range(3):
        t = [torch.from_numpy(a.copy()).cuda() for a in (x, y, z, h)]
        # (every buffer that takes part in the exchange offers n elements of the coordinates' size: float32 masses next to
        #  float64 coordinates sit in the first half of a tensor of 2 n)
        mbuf = torch.zeros(n * (rb // mass_bits), dtype=torch.float64 if mass_bits == 64 else torch.float32, device="cuda")
        mbuf[:n] = torch.from_numpy(m.copy()).cuda()
        keys = torch.zeros(n, dtype=torch.int64 if kb == 64 else torch.int32, device="cuda")
        scratch = [torch.empty_like(t[0]) for _ in range(3)]
        ident = torch.arange(n, dtype=t[0].dtype, device="cuda")
        keys, xd, yd, zd, hd, md, scratch, (ident,) = dom.sync_grav(keys, *t, mbuf[:n], scratch, [ident])
        v = dom.view()
        order = ident.cpu().numpy().astype(np.int64)
        assert np.array_equal(md.cpu().numpy(), m[order]) and np.array_equal(xd.cpu().numpy(), x[order])
        assert v.expansion_centers
        got = dom.fetch(v.expansion_centers, 4 * v.num_focus_nodes, rdt).reshape(-1, 4)
        want = expect(v, x[order], y[order], z[order], m[order])
        assert np.array_equal(got[:, :3], want[:, :3]), (s, "centres", int((got[:, :3] != want[:, :3]).any(1).sum()))
        assert np.array_equal(got[:, 3], want[:, 3]), (s, "MAC radii", int((got[:, 3] != want[:, 3]).sum()),
                                                        float(np.abs(got[:, 3] / want[:, 3] - 1).max()))
        # the root: the centre of mass of the whole cloud (to rounding), a positive MAC radius
        com = (m[:, None].astype(np.float64) * np.stack([x, y, z], 1)).sum(0) / m.sum(dtype=np.float64)
        assert np.allclose(got[0, :3], com, rtol=1e-4 if rb == 32 else 1e-10, atol=1e-6) and got[0, 3] > 0
        # the same tree as a plain sync of the same input
        t2 = [torch.from_numpy(a.copy()).cuda() for a in (x, y, z, h)]
        k2 = torch.zeros_like(keys)
        same_tree.sync(k2, *t2, [torch.empty_like(t2[0]) for _ in range(3)])
        v2 = same_tree.view()
        assert v2.num_focus_leaves == v.num_focus_leaves and not v2.expansion_centers
        assert np.array_equal(same_tree.fetch(v2.focus_leaves, v2.num_focus_leaves + 1, kdt),
                              dom.fetch(v.focus_leaves, v.num_focus_leaves + 1, kdt))
        # Domain::updateExpansionCenters with other masses on the synced arrays
        m2 = (m[order] * rng.uniform(0.5, 1.5, n)).astype(mdt)
        m2d = torch.from_numpy(m2).cuda()
        dom.update_expansion_centers(xd, yd, zd, m2d)
        got2 = dom.fetch(dom.view().expansion_centers, 4 * v.num_focus_nodes, rdt).reshape(-1, 4)
        assert np.array_equal(got2, expect(v, x[order], y[order], z[order], m2)), s
        # move
        x = np.clip(x + rng.normal(0, 0.01, n).astype(rdt), -0.99, 0.99).astype(rdt)
        z = np.clip(z + rng.normal(0, 0.01, n).astype(rdt), -0.49, 0.49).astype(rdt)
    with pytest.raises(Exception):
        Domain(hip, HILBERT, kb, rb, 64, 16, theta, cstone_amd.make_cbox(lim, bc)).update_expansion_centers(xd, yd, zd, md)
