"""The incremental re-sort inside cstone_hip_domain_sync (csrc/resort.hpp): a sync that starts from the arrays the previous
sync returned orders the particles leaf by leaf (stayers in LDS, movers through bins) instead of radix-sorting all keys.
The reference always sorts from scratch (sortByKeyGpu, primitives_gpu.cu:305-353, via sfc_sorter.hpp), so the re-sort must
give exactly the stable sort: every test runs the same time-stepping loop through a domain that may re-sort and through one
that may not (cstone_hip_domain_set_sort_mode: sorted from scratch) and compares everything bit for bit, next to the oracle's keys under the domain's box."""
import os

import numpy as np
import pytest


_NUM_SCRATCH = [1]  # scratch arrays a _Stepper hands to its syncs (set by the fixture below)


@pytest.fixture(autouse=True, params=["auto", "one-per-lane", "two-per-lane", "auto-fields", "one-per-lane-fields",
                                      "two-per-lane-fields"])
def leaf_pass_flavour(request, monkeypatch):
    """every test of this module runs with the leaf pass choosing its flavour itself (by the fill of the leaves) and with
    either flavour of leafSortWaveKernel forced (CSTONE_RESORT_PAIRS, csrc/resort.hip) -- and each of these once with ONE
    scratch array (the leaf pass orders keys and indices, the fields follow in gather passes) and once with FOUR and
    CSTONE_FUSED_LEAF_PASS (the field-carrying leaf pass: x, y, z, h move with the keys, the halo radii come from the
    maxima it folds)"""
    flavour = request.param.replace("-fields", "")
    if flavour != "auto":
        monkeypatch.setenv("CSTONE_RESORT_PAIRS", "0" if flavour == "one-per-lane" else "1")
    _NUM_SCRATCH[0] = 4 if request.param.endswith("-fields") else 1
    if request.param.endswith("-fields"):
        monkeypatch.setenv("CSTONE_FUSED_LEAF_PASS", "1")
    yield
    _NUM_SCRATCH[0] = 1


class _Stepper:
    """a client's time-stepping loop: the arrays a sync returns are moved in place and handed to the next sync"""

    def __init__(self, hip, kb, rb, bucket_focus, curve, bc, n, seed, allow_resort):
        import torch

        import cstone_amd
        from cstone_amd.domain import Domain

        rng = np.random.default_rng(seed)
        self.hip, self.kb, self.allow = hip, kb, allow_resort
        rdt = torch.float64 if rb == 64 else torch.float32
        kdt = torch.int64 if kb == 64 else torch.int32
        box = cstone_amd.make_cbox([0, 1, 0, 1, 0, 1], bc)
        self.dom = Domain(hip, curve, kb, rb, max(bucket_focus, 64) * 4, bucket_focus, 0.5, box)
        centers = rng.uniform(0.2, 0.8, (5, 3))
        pos = np.where(rng.uniform(size=(n, 1)) < 0.5, rng.uniform(0, 1, (n, 3)),
                       centers[rng.integers(0, 5, n)] + rng.normal(0, 0.03, (n, 3)))
        pos = np.clip(pos, 0.0, 1.0 - 1e-6)
        self.x, self.y, self.z = [torch.from_numpy(np.ascontiguousarray(pos[:, d])).to(rdt).cuda() for d in range(3)]
        self.h = torch.from_numpy(rng.uniform(0.001, 0.01, n)).to(rdt).cuda()
        self.ident = torch.arange(n, dtype=rdt, device="cuda")  # follows its particle through every sync
        self.keys = torch.zeros(n, dtype=kdt, device="cuda")
        self.num_scratch = _NUM_SCRATCH[0]
        self.scratch = self._new_scratch()

    def _new_scratch(self):
        import torch

        return torch.empty_like(self.x) if self.num_scratch == 1 else [torch.empty_like(self.x) for _ in range(self.num_scratch)]

    def sync(self):
        import torch

        # (cstone_hip_domain_set_sort_mode: the incremental re-sort, or the radix sort from scratch -- identical results;
        #  allow = None: the test has chosen the mode itself)
        if self.allow is not None:
            self.dom.set_sort_mode(self.dom.SORT_INCREMENTAL if self.allow else self.dom.SORT_FROM_SCRATCH)
        self.keys, self.x, self.y, self.z, self.h, self.scratch, (self.ident,) = self.dom.sync(
            self.keys, self.x, self.y, self.z, self.h, self.scratch, [self.ident])
        self.hip.sync()
        m = self.x.numel()
        if (self.scratch if self.num_scratch == 1 else self.scratch[0]).numel() != m:
            # particles were removed: the client shrinks its arrays to the new size (views of the old buffers would do
            # as well; fresh ones keep every array 16-byte aligned for this test)
            self.keys, self.x, self.y, self.z, self.h, self.ident = [t.clone() for t in (
                self.keys, self.x, self.y, self.z, self.h, self.ident)]
            self.scratch = self._new_scratch()
        return dict(keys=self.keys.cpu().numpy(), x=self.x.cpu().numpy(), y=self.y.cpu().numpy(),
                    z=self.z.cpu().numpy(), h=self.h.cpu().numpy(), ident=self.ident.cpu().numpy(), view=self.dom.view())

    def move(self, kind, mrng):
        """in place, on the arrays the last sync returned"""
        import torch

        coords = (self.x, self.y, self.z)
        m = self.x.numel()

        def put(sel, values, a):
            a[sel] = torch.from_numpy(values).to(a.dtype).cuda()

        if kind == "none":
            return
        # nobody leaves the extents the particles have now: with open boundaries the box of the next sync stays what it
        # is (a box that changes re-encodes every key and takes the regular path, covered by test_domain.py)
        ext = [(float(a.min()), float(a.max())) for a in coords]
        if kind == "jitter":  # everybody, by a fraction of a leaf
            for a, (lo, hi) in zip(coords, ext):
                a.add_(torch.from_numpy(mrng.normal(0, 2e-4, m)).to(a.dtype).cuda()).clamp_(lo, hi)
        elif kind in ("few", "many"):  # 2 % (or every second particle: too many for the re-sort) jump anywhere
            sel = torch.from_numpy(mrng.choice(m, max(1, m // (50 if kind == "few" else 2)), replace=False)).cuda()
            for a, (lo, hi) in zip(coords, ext):
                put(sel, mrng.uniform(lo, hi, sel.numel()), a)
                a.clamp_(lo, hi)
        elif kind == "pairs":  # 2 % of the particles onto the position of another one: equal keys inside ordinary leaves
            sel = torch.from_numpy(mrng.choice(m, max(2, m // 50), replace=False)).cuda()
            onto = torch.from_numpy(mrng.choice(m, sel.numel(), replace=True)).cuda()
            for a in coords:
                a[sel] = a[onto]
        elif kind == "collapse":  # a tenth of the particles onto three points: equal keys, overfull leaves
            sel = torch.from_numpy(mrng.choice(m, m // 10, replace=False)).cuda()
            pts = mrng.uniform(0.1, 0.9, (3, 3))
            which = mrng.integers(0, 3, sel.numel())
            for d, a in enumerate(coords):
                put(sel, pts[which, d], a)
        elif kind == "remove":  # flag 1 % with the remove marker (domain.hpp: particles with removeKey leave)
            sel = torch.from_numpy(mrng.choice(m, max(1, m // 100), replace=False)).cuda()
            self.keys[sel] = torch.iinfo(torch.int64).min if self.kb == 64 else (1 << 30)  # bit pattern of endKey
        elif kind == "shuffle":  # the client reorders its arrays behind the domain's back
            perm = torch.from_numpy(mrng.permutation(m)).cuda()
            self.x, self.y, self.z, self.h, self.ident = [t[perm].contiguous() for t in (
                self.x, self.y, self.z, self.h, self.ident)]
        else:
            raise ValueError(kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kb,rb,bucket_focus,curve,bc,always_count,n", [
    (64, 64, 64, 1, (0, 0, 0), False, 120_000),
    (64, 64, 64, 1, (0, 0, 0), True, 120_000),  # every tile of the leaf pass through the counting path, quiet or not
    (64, 64, 16, 1, (1, 1, 1), False, 120_000),
    (64, 32, 100, 1, (0, 1, 2), False, 120_000),
    (32, 32, 64, 0, (0, 0, 0), False, 120_000),
    (32, 32, 64, 0, (0, 0, 0), True, 120_000),
    (32, 64, 200, 1, (1, 1, 1), False, 120_000),
    (64, 64, 1000, 1, (0, 0, 0), False, 120_000),  # buckets beyond the leaf pass: never re-sorted, same results
    # a handful of leaves at levels 1-2: too wide for the packed (key, slot) entries of the counting path
    (64, 64, 64, 1, (0, 0, 0), True, 700),
    (32, 32, 64, 1, (1, 1, 1), True, 700),
    (64, 64, 64, 0, (0, 0, 0), False, 700),
])
def test_resort_equals_full_sort_over_a_time_stepping_loop(hip, oracle, monkeypatch, kb, rb, bucket_focus, curve, bc,
                                                           always_count, n):
    from oracle.oracle import Box

    if always_count:
        monkeypatch.setenv("CSTONE_RESORT_COUNT", "1")

    seed = 7 + kb + bucket_focus
    sa_, sb_ = (_Stepper(hip, kb, rb, bucket_focus, curve, bc, n, seed, allow) for allow in (True, False))
    dom_a, dom_b = sa_.dom, sb_.dom
    kdt = np.uint64 if kb == 64 else np.uint32
    script = ["none", "none", "jitter", "few", "pairs", "jitter", "remove", "pairs", "few", "many", "none", "collapse",
              "jitter", "none", "few", "none"]
    for step, kind in enumerate(script):
        if step:
            sa_.move(kind, np.random.default_rng(1000 + step))
            sb_.move(kind, np.random.default_rng(1000 + step))
        a, b = sa_.sync(), sb_.sync()
        va, vb = a["view"], b["view"]
        assert (va.end_index, va.num_focus_leaves) == (vb.end_index, vb.num_focus_leaves), (step, kind)
        for f in ("keys", "x", "y", "z", "h", "ident"):
            assert np.array_equal(a[f], b[f]), (step, kind, f)
        m, L = va.end_index, va.num_focus_leaves
        assert np.array_equal(dom_a.fetch(va.sfc_order, m, np.uint32), dom_b.fetch(vb.sfc_order, m, np.uint32)), step
        assert np.array_equal(dom_a.fetch(va.layout, L + 1, np.uint32), dom_b.fetch(vb.layout, L + 1, np.uint32)), step
        assert np.array_equal(dom_a.fetch(va.focus_leaves, L + 1, kdt), dom_b.fetch(vb.focus_leaves, L + 1, kdt)), step
        # the halo radii (2 * max h per leaf): folded by the leaf pass on one side, by the gather of h on the other
        ra_, rb_ = dom_a.fetch(va.halo_radii, L, np.float32), dom_b.fetch(vb.halo_radii, L, np.float32)
        assert np.array_equal(ra_, rb_), (step, kind, np.nonzero(ra_ != rb_)[0][:5])
        # and against the oracle: the keys of the returned coordinates under the domain's box, ascending
        want = oracle.compute_sfc_keys(curve, kb, a["x"], a["y"], a["z"], Box(list(va.box.lim), bc))
        assert np.array_equal(a["keys"].view(kdt), want), (step, kind)
        assert np.all(want[1:] >= want[:-1]), (step, kind)
    sa, sb = dom_a.stats(), dom_b.stats()
    assert sb["resorts"] == 0
    if bucket_focus < 256:  # a leaf must stay below 256 slots (RESORT_LEAF_CAP)
        # "none", "jitter", "few", "remove" steps re-sort; "many" and "collapse" are given up (and back off four syncs)
        if n > 1000:
            assert sa["resorts"] >= 5 and sa["resort_fallbacks"] >= 1, sa
        else:
            assert sa["resorts"] >= 3, sa  # (a few hundred particles: the box follows nearly every move)
    else:
        assert sa["resorts"] == 0, sa


@pytest.mark.gpu
def test_resort_ignores_what_the_caller_did_to_the_arrays(hip, oracle):
    """the arrays of the previous sync are only a hint: a caller that hands in shuffled arrays (or new particles in the
    old buffers) gets the same result as from a fresh domain -- every particle is then a mover, the attempt is given up"""
    from oracle.oracle import Box

    st_ = _Stepper(hip, 64, 64, 64, 1, (0, 0, 0), 80_000, 3, True)
    dom = st_.dom
    st_.sync()
    a = st_.sync()
    assert dom.stats()["resorts"] == 1 and dom.stats()["last_movers"] == 0
    st_.move("shuffle", np.random.default_rng(5))
    b = st_.sync()
    st = dom.stats()
    assert st["resort_fallbacks"] == 1, st
    want = oracle.compute_sfc_keys(1, 64, b["x"], b["y"], b["z"], Box(list(b["view"].box.lim), (0, 0, 0)))
    assert np.array_equal(b["keys"].view(np.uint64), want) and np.all(want[1:] >= want[:-1])
    assert np.array_equal(np.sort(b["ident"]), np.sort(a["ident"]))


@pytest.mark.gpu
def test_resort_at_three_million_particles(hip, oracle):
    """many workgroups of the leaf pass, movers in the hundred thousands, 63-bit keys"""
    from oracle.oracle import Box

    st_ = _Stepper(hip, 64, 64, 64, 1, (0, 0, 0), 3_000_000, 11, True)
    dom = st_.dom
    st_.sync()
    for step, kind in enumerate(["none", "jitter", "few", "jitter"]):
        st_.move(kind, np.random.default_rng(50 + step))
        a = st_.sync()
        want = oracle.compute_sfc_keys(1, 64, a["x"], a["y"], a["z"], Box(list(a["view"].box.lim), (0, 0, 0)))
        assert np.array_equal(a["keys"].view(np.uint64), want), kind
        assert np.all(want[1:] >= want[:-1]), kind
        assert np.unique(a["ident"]).size == a["ident"].size
    st = dom.stats()
    assert st["resorts"] == 4 and st["resort_fallbacks"] == 0, st


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 7, 63, 64, 65, 1000])
def test_resort_with_a_handful_of_particles(hip, oracle, n):
    """one leaf, one 64-position word of the leaf table, fewer particles than a wave"""
    from oracle.oracle import Box

    st_ = _Stepper(hip, 64, 64, 64, 1, (1, 1, 1), n, 21, True)
    for step, kind in enumerate(["none", "none", "jitter", "few", "none"]):
        if step:
            st_.move(kind, np.random.default_rng(70 + step))
        a = st_.sync()
        want = oracle.compute_sfc_keys(1, 64, a["x"], a["y"], a["z"], Box(list(a["view"].box.lim), (1, 1, 1)))
        assert np.array_equal(a["keys"].view(np.uint64), want) and np.all(want[1:] >= want[:-1]), (n, kind)
        assert np.array_equal(np.sort(a["ident"]), np.arange(n)), (n, kind)
    assert st_.dom.stats()["resorts"] >= 2, st_.dom.stats()


@pytest.mark.gpu
def test_unaligned_arrays_take_the_regular_path(hip, oracle):
    """coordinate arrays that do not start on a 16-byte boundary cannot go through the vector encode kernels: the sync
    falls back to the plain encode and the radix sort, same results"""
    import torch
    from oracle.oracle import Box

    st_ = _Stepper(hip, 64, 64, 64, 1, (1, 1, 1), 50_001, 22, True)
    st_.sync()
    # views shifted by one element (8 bytes): the client dropped its first particle
    st_.x, st_.y, st_.z, st_.h, st_.ident = [t[1:] for t in (st_.x, st_.y, st_.z, st_.h, st_.ident)]
    st_.keys, st_.scratch = st_.keys[1:], torch.empty(50_001, dtype=st_.x.dtype, device="cuda")[1:]
    from cstone_amd.domain import Domain
    import cstone_amd

    # a new domain: the old one insists on the size of its last sync
    st_.dom = Domain(hip, 1, 64, 64, 256, 64, 0.5, cstone_amd.make_cbox([0, 1, 0, 1, 0, 1], (1, 1, 1)))
    for _ in range(3):
        a = st_.sync()
        want = oracle.compute_sfc_keys(1, 64, a["x"], a["y"], a["z"], Box(list(a["view"].box.lim), (1, 1, 1)))
        assert np.array_equal(a["keys"].view(np.uint64), want) and np.all(want[1:] >= want[:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_resort_random_walk(hip, oracle, seed):
    """forty syncs with moves drawn at random (quiet, jitter, jumps, equal keys, removals, shuffles, half of the cloud
    jumping, collapses), a bucket size drawn at random: the re-sorting domain and the one that never re-sorts agree on
    everything after every sync"""
    rng = np.random.default_rng(900 + seed)
    kb = int(rng.choice([32, 64]))
    bucket_focus = int(rng.choice([8, 64, 150]))
    bc = tuple(int(v) for v in rng.choice([0, 1], 3))
    n = int(rng.choice([20_000, 90_000]))
    sa_, sb_ = (_Stepper(hip, kb, 64, bucket_focus, 1, bc, n, 31 + seed, allow) for allow in (True, False))
    kinds = ["none", "jitter", "jitter", "few", "few", "pairs", "remove", "shuffle", "many", "collapse"]
    for step in range(40):
        kind = "none" if step == 0 else str(rng.choice(kinds))
        if step:
            sa_.move(kind, np.random.default_rng(5000 + 100 * seed + step))
            sb_.move(kind, np.random.default_rng(5000 + 100 * seed + step))
        a, b = sa_.sync(), sb_.sync()
        va, vb = a["view"], b["view"]
        assert (va.end_index, va.num_focus_leaves) == (vb.end_index, vb.num_focus_leaves), (step, kind)
        for f in ("keys", "x", "y", "z", "h", "ident"):
            assert np.array_equal(a[f], b[f]), (step, kind, f)
        m = va.end_index
        assert np.array_equal(sa_.dom.fetch(va.sfc_order, m, np.uint32), sb_.dom.fetch(vb.sfc_order, m, np.uint32)), step
    # (how often it re-sorted depends on the draw: collapses leave overfull leaves behind; the other tests pin that down)
    assert sb_.dom.stats()["resorts"] == 0 and sa_.dom.stats()["resorts"] >= 1, sa_.dom.stats()


@pytest.mark.gpu
def test_full_leaves_everywhere(hip, oracle):
    """a regular 64^3 lattice with bucket 64: every leaf holds exactly 64 particles, a tile of 64 leaves has 4096 old
    slots -- more than the quiet instantiation of the leaf pass takes, so the other one orders them although nothing
    moves; then a jitter, which sends arrivals into full leaves"""
    import torch
    from oracle.oracle import Box

    st_ = _Stepper(hip, 64, 64, 64, 1, (1, 1, 1), 64 ** 3, 41, True)
    g = (np.arange(64) + 0.5) / 64
    gx, gy, gz = np.meshgrid(g, g, g, indexing="ij")
    st_.x, st_.y, st_.z = [torch.from_numpy(a.ravel().copy()).cuda() for a in (gx, gy, gz)]
    for step, kind in enumerate(["none", "none", "none", "jitter", "none", "few", "none"]):
        if step:
            st_.move(kind, np.random.default_rng(80 + step))
        a = st_.sync()
        want = oracle.compute_sfc_keys(1, 64, a["x"], a["y"], a["z"], Box(list(a["view"].box.lim), (1, 1, 1)))
        assert np.array_equal(a["keys"].view(np.uint64), want) and np.all(want[1:] >= want[:-1]), (step, kind)
        assert np.array_equal(np.sort(a["ident"]), np.arange(64 ** 3)), (step, kind)
        if step == 2:
            counts = st_.dom.fetch(a["view"].focus_leaf_counts, a["view"].num_focus_leaves, np.uint32)
            assert counts.min() == 64 and counts.max() == 64  # the premise of this test
    assert st_.dom.stats()["resorts"] >= 3, st_.dom.stats()


@pytest.mark.gpu
def test_open_box_whose_outermost_particles_move(hip, oracle, monkeypatch):
    """open boundaries, the outermost particles move with every step: the box of limitBoxShrinking (R/sfc/box.hpp:415-431)
    changes with every sync, keys computed with the previous box are worthless.  After ONE such sync the domain measures
    the extents before it encodes (no second speculative encode is thrown away) and goes back to speculating after a sync
    whose box stayed; the results equal those of a domain that never speculates (CSTONE_NO_SPECULATIVE_BOX) and the
    oracle's keys under the domain's box"""
    from oracle.oracle import Box

    kb, rb, bucket_focus, curve, bc, n = 64, 64, 64, 1, (0, 0, 0), 150_000
    sa_, sb_ = (_Stepper(hip, kb, rb, bucket_focus, curve, bc, n, 31, True) for _ in range(2))
    script = ["none", "expand", "expand", "expand", "none", "none", "expand", "none", "jitter", "none"]
    boxes = []
    for step, kind in enumerate(script):
        for s_ in (sa_, sb_):
            if kind == "expand":  # everybody drifts outwards a little: all six extents move
                for a in (s_.x, s_.y, s_.z):
                    a.sub_(0.5).mul_(1.001).add_(0.5)
            elif step:
                s_.move(kind, np.random.default_rng(2000 + step))
        a = sa_.sync()
        monkeypatch.setenv("CSTONE_NO_SPECULATIVE_BOX", "1")
        b = sb_.sync()
        monkeypatch.delenv("CSTONE_NO_SPECULATIVE_BOX")
        va, vb = a["view"], b["view"]
        assert list(va.box.lim) == list(vb.box.lim), (step, kind)
        for f in ("keys", "x", "y", "z", "h", "ident"):
            assert np.array_equal(a[f], b[f]), (step, kind, f)
        L = va.num_focus_leaves
        assert L == vb.num_focus_leaves
        assert np.array_equal(sa_.dom.fetch(va.layout, L + 1, np.uint32), sb_.dom.fetch(vb.layout, L + 1, np.uint32)), step
        want = oracle.compute_sfc_keys(curve, kb, a["x"], a["y"], a["z"], Box(list(va.box.lim), bc))
        assert np.array_equal(a["keys"].view(np.uint64), want), (step, kind)
        boxes.append(list(va.box.lim))
    assert boxes[1] != boxes[0] and boxes[2] != boxes[1] and boxes[5] == boxes[4]
    sa, sb = sa_.dom.stats(), sb_.dom.stats()
    # speculative encodes thrown away: the first expanding step and the one after the quiet stretch, not every one of the four
    assert sb["box_redos"] == 0 and 1 <= sa["box_redos"] <= 2, (sa, sb)
    # the quiet steps behind an unchanged box are re-sorted again in both domains
    assert sa["resorts"] >= 2 and sb["resorts"] >= 2, (sa, sb)


@pytest.mark.gpu
def test_sort_mode_setters_give_the_same_result(hip, oracle):
    """cstone_hip_domain_set_sort_mode / _set_speculative_box: the client-side switches behind the environment variables
    of the experiments.  Three domains walk the same loop -- incremental (default), from scratch, all digits without box
    speculation -- and agree bit for bit; only the first one re-sorts"""
    doms = [_Stepper(hip, 64, 64, 64, 1, (0, 0, 0), 90_000, 5, None) for _ in range(3)]
    doms[1].dom.set_sort_mode(doms[1].dom.SORT_FROM_SCRATCH)
    doms[2].dom.set_sort_mode(doms[2].dom.SORT_ALL_DIGITS)
    doms[2].dom.set_speculative_box(False)
    for step, kind in enumerate(["none", "jitter", "few", "none", "jitter"]):
        outs = []
        for d in doms:
            if step:
                d.move(kind, np.random.default_rng(300 + step))
            outs.append(d.sync())
        for o in outs[1:]:
            for f in ("keys", "x", "y", "z", "h", "ident"):
                assert np.array_equal(outs[0][f], o[f]), (step, kind, f)
    st = [d.dom.stats() for d in doms]
    assert st[0]["resorts"] >= 3 and st[1]["resorts"] == 0 and st[2]["resorts"] == 0, st
