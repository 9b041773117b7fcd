"""Golden fixtures produced by the reference's own CPU code (tests/golden/make_golden.py).
CPU: the oracle must reproduce every array.  GPU (-m gpu): the HIP library must reproduce every array."""
import glob
import os

import numpy as np
import pytest

from helpers import Box
from oracle.oracle import HILBERT, MORTON

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_k*_f*.npz")))
CURVES = ((MORTON, "morton"), (HILBERT, "hilbert"))


def _load(path):
    d = np.load(path)
    name = os.path.basename(path)
    kb, rb = int(name[5:7]), int(name[9:11])
    return d, kb, rb


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_reproduces_reference_goldens(oracle, path):
    d, kb, rb = _load(path)
    box = Box(d["lim"])
    n = int(d["n"])
    x, y, z = d["x"], d["y"], d["z"]
    for curve, cn in CURVES:
        keys = oracle.compute_sfc_keys(curve, kb, x, y, z, box)
        assert np.array_equal(keys, d[f"{cn}_keys"])
        ks, order = oracle.sort_pairs(keys, np.arange(n))
        assert np.array_equal(order, d[f"{cn}_order"])
        tree = np.array([0, 1 << (3 * (10 if kb == 32 else 21))], dtype=keys.dtype)
        counts = np.array([n], dtype=np.uint32)
        for it in range(int(d[f"{cn}_iters"])):
            tree, counts, conv = oracle.update_octree(ks, 16, tree, counts)
            assert np.array_equal(tree, d[f"{cn}_tree_it{it}"]) and np.array_equal(counts, d[f"{cn}_counts_it{it}"])
        assert conv
        o = oracle.build_octree(tree)
        for k in ("prefixes", "child_offsets", "parents", "level_range", "internal_to_leaf", "leaf_to_internal"):
            assert np.array_equal(o[k], d[f"{cn}_oct_{k}"]), k
        if curve == HILBERT:
            nl = tree.size - 1
            for bcn, bc in (("open", (0, 0, 0)), ("pbc", (1, 1, 1))):
                for (f, l) in ((0, nl // 4), (nl // 4, 3 * nl // 4)):
                    got = oracle.find_halos(HILBERT, o, tree, d["halo_radii"], Box(d["lim"], bc), f, l, rb)
                    assert np.array_equal(got, d[f"halo_flags_{bcn}_{f}_{l}"])
            cen, siz = oracle.node_centers(HILBERT, o["prefixes"], box, rb)
            assert np.array_equal(cen, d["centers"]) and np.array_equal(siz, d["sizes"])
            layout = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
            xs, ys, zs = x[order], y[order], z[order]
            for bcn, bc in (("open", (0, 0, 0)), ("pbc", (1, 0, 1))):
                nidx, nc = oracle.find_neighbors(xs, ys, zs, d["h_sorted"], 0, n, Box(d["lim"], bc), o, layout, cen,
                                                 siz, 32)
                assert np.array_equal(nc, d[f"nc_{bcn}"]) and np.array_equal(nidx, d[f"nidx_{bcn}"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_hip_reproduces_reference_goldens(hip, path):
    import cstone_amd
    from test_gpu_parity import dev, host

    d, kb, rb = _load(path)
    n = int(d["n"])
    x, y, z = d["x"], d["y"], d["z"]
    import torch

    for curve, cn in CURVES:
        cb = cstone_amd.make_cbox(d["lim"])
        keys = hip.compute_sfc_keys(curve, kb, dev(x), dev(y), dev(z), cb)
        assert np.array_equal(host(keys), d[f"{cn}_keys"])
        order = torch.arange(n, dtype=torch.int32, device="cuda")
        hip.sort_pairs(keys, order)
        assert np.array_equal(host(order), d[f"{cn}_order"])
        cap = 8 * n
        tb = torch.zeros(cap + 1, dtype=keys.dtype, device="cuda")
        cbuf = torch.zeros(cap, dtype=torch.int32, device="cuda")
        tb[:2] = dev(np.array([0, 1 << (3 * (10 if kb == 32 else 21))], dtype=d[f"{cn}_keys"].dtype))
        cbuf[:1] = n
        nl = 1
        for it in range(int(d[f"{cn}_iters"])):
            nl, conv = hip.update_octree(keys, 16, tb, cbuf, nl)
            assert np.array_equal(host(tb[:nl + 1]), d[f"{cn}_tree_it{it}"])
            assert np.array_equal(host(cbuf[:nl]), d[f"{cn}_counts_it{it}"])
        assert conv
        tree = tb[:nl + 1].clone()
        o = hip.build_octree(tree)
        for k in ("prefixes", "level_range", "internal_to_leaf", "leaf_to_internal"):
            assert np.array_equal(host(o[k], unsigned=(k == "prefixes")), d[f"{cn}_oct_{k}"]), k
        nn = o["num_nodes"]
        assert np.array_equal(host(o["child_offsets"], False)[:nn], d[f"{cn}_oct_child_offsets"][:nn])
        npar = d[f"{cn}_oct_parents"].size
        assert np.array_equal(host(o["parents"], False)[:npar], d[f"{cn}_oct_parents"])
        if curve == HILBERT:
            for bcn, bc in (("open", (0, 0, 0)), ("pbc", (1, 1, 1))):
                for (f, l) in ((0, nl // 4), (nl // 4, 3 * nl // 4)):
                    got = hip.find_halos(HILBERT, o, tree, dev(d["halo_radii"]), cstone_amd.make_cbox(d["lim"], bc),
                                         f, l, rb)
                    assert np.array_equal(host(got, False), d[f"halo_flags_{bcn}_{f}_{l}"])
            cen, siz = hip.node_centers(HILBERT, o["prefixes"], cb, rb)
            assert np.array_equal(host(cen, False), d["centers"]) and np.array_equal(host(siz, False), d["sizes"])
            layout = np.concatenate([[0], np.cumsum(d[f"{cn}_counts_it{int(d[cn + '_iters']) - 1}"])]).astype(np.uint32)
            perm = d[f"{cn}_order"]
            for bcn, bc in (("open", (0, 0, 0)), ("pbc", (1, 0, 1))):
                nidx, nc = hip.find_neighbors(dev(x[perm]), dev(y[perm]), dev(z[perm]), dev(d["h_sorted"]), 0, n,
                                              cstone_amd.make_cbox(d["lim"], bc), o, dev(layout), cen, siz, 32)
                nc, nidx = host(nc), host(nidx)
                assert np.array_equal(nc, d[f"nc_{bcn}"])
                mask = np.arange(32)[None, :] < np.minimum(nc, 32)[:, None]
                assert np.array_equal(nidx[mask], d[f"nidx_{bcn}"][mask])
    hip.sync()
