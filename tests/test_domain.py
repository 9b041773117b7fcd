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
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_domain_sync_matches_reference(hip, path):
    import torch

    import cstone_amd
    from cstone_amd.domain import Domain

    d = np.load(path)
    kb = int(d["key_bits"]) if "key_bits" in d else 64
    rb = int(d["real_bits"]) if "real_bits" in d else 64
    kdt, ksigned = (np.uint64, np.int64) if kb == 64 else (np.uint32, np.int32)
    box = cstone_amd.make_cbox(d["lim"], d["bc"])
    dom = Domain(hip, cstone_amd.HILBERT, kb, rb, int(d["bucket"]), int(d["bucket_focus"]), 0.5, box)
    for s in range(int(d["steps"])):
        x, y, z, h = [torch.from_numpy(d[f"in{s}_{c}"].copy()).cuda() for c in "xyzh"]
        assert x.element_size() * 8 == rb
        n = x.numel()
        kin = d[f"in{s}_keys"] if f"in{s}_keys" in d else np.zeros(n, kdt)
        keys = torch.from_numpy(kin.view(ksigned).copy()).cuda()
        scratch = torch.empty_like(x)
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
