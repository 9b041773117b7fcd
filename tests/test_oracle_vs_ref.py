"""CPU tests: the oracle restatement against the reference's own CPU code (oracle/_ref) on randomized inputs.
Skipped where the reference build is not available (the GPU box receives the prebuilt library; a container
without /root/reference and without oracle/_ref skips)."""
import numpy as np
import pytest

from helpers import Box, end_key, key_dtype, random_cloud, real_dtype
from oracle.oracle import HILBERT, MORTON


@pytest.mark.parametrize("kb", [32, 64])
@pytest.mark.parametrize("rb", [32, 64])
@pytest.mark.parametrize("kind", ["uniform", "clustered"])
def test_pipeline_matches_reference(oracle, reference, kb, rb, kind):
    box = Box([-1.3, 2.1, 0.2, 0.9, -5, 7])
    n = 20000
    x, y, z = random_cloud(n, box, rb, seed=kb * rb, kind=kind)
    for curve in (MORTON, HILBERT):
        ko = oracle.compute_sfc_keys(curve, kb, x, y, z, box)
        kr = reference.compute_sfc_keys(curve, kb, x, y, z, box)
        assert np.array_equal(ko, kr)
        so, vo = oracle.sort_pairs(ko, np.arange(n))
        sr, vr = reference.sort_pairs(kr, np.arange(n))
        assert np.array_equal(so, sr) and np.array_equal(vo, vr)
        for bucket in (1, 16, 64):
            to, co = oracle.compute_octree(so, bucket)
            tr, cr = reference.compute_octree(sr, bucket)
            assert np.array_equal(to, tr) and np.array_equal(co, cr)
        oo, orr = oracle.build_octree(to), reference.build_octree(tr)
        for k in oo:
            assert np.array_equal(oo[k], orr[k]), k
        lc = np.random.default_rng(1).integers(0, 2**31, to.size - 1, dtype=np.uint32)
        assert np.array_equal(oracle.upsweep_counts(oo, lc), reference.upsweep_counts(orr, lc))
        if curve == HILBERT:
            nl = to.size - 1
            radii = np.random.default_rng(2).uniform(0.005, 0.2, nl).astype(np.float32)
            for bc in ((0, 0, 0), (1, 1, 1), (0, 1, 0)):
                pb = Box(box.lim, bc)
                for f, l in ((0, nl // 3), (nl // 3, nl), (0, nl)):
                    assert np.array_equal(oracle.find_halos(curve, oo, to, radii, pb, f, l, rb),
                                          reference.find_halos(curve, orr, tr, radii, pb, f, l, rb))
            c1, s1 = oracle.node_centers(curve, oo["prefixes"], box, rb)
            c2, s2 = reference.node_centers(curve, orr["prefixes"], box, rb)
            assert np.array_equal(c1, c2) and np.array_equal(s1, s2)


@pytest.mark.parametrize("rb", [32, 64])
@pytest.mark.parametrize("bc", [(0, 0, 0), (1, 1, 1), (1, 0, 0)])
def test_neighbors_match_reference(oracle, reference, rb, bc):
    box = Box([0, 1, 0, 1, 0, 1], bc)
    n = 6000
    x, y, z = random_cloud(n, box, rb, seed=77, kind="clustered")
    keys = oracle.compute_sfc_keys(HILBERT, 64, x, y, z, box)
    ks, order = oracle.sort_pairs(keys, np.arange(n))
    x, y, z = x[order], y[order], z[order]
    h = (0.04 * np.random.default_rng(3).uniform(0.5, 1.5, n)).astype(real_dtype(rb))
    tree, counts = oracle.compute_octree(ks, 16)
    o = oracle.build_octree(tree)
    layout = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    cen, siz = oracle.node_centers(HILBERT, o["prefixes"], box, rb)
    n1, c1 = oracle.find_neighbors(x, y, z, h, 10, n - 10, box, o, layout, cen, siz, 48)
    n2, c2 = reference.find_neighbors(x, y, z, h, 10, n - 10, box, o, layout, cen, siz, 48)
    assert np.array_equal(c1, c2) and np.array_equal(n1, n2)


@pytest.mark.parametrize("kb", [32, 64])
def test_stepwise_updates_and_spanning_tree(oracle, reference, kb):
    rng = np.random.default_rng(kb)
    keys = np.sort(rng.integers(0, end_key(kb), 50000, dtype=np.uint64)).astype(key_dtype(kb))
    tree = np.array([0, end_key(kb)], dtype=key_dtype(kb))
    counts = np.array([keys.size], dtype=np.uint32)
    for _ in range(10):
        ops_o, conv_o = oracle.node_ops(tree, counts, 8)
        ops_r, conv_r = reference.node_ops(tree, counts, 8)
        assert np.array_equal(ops_o, ops_r) and conv_o == conv_r
        t1, c1, k1 = oracle.update_octree(keys, 8, tree, counts)
        t2, c2, k2 = reference.update_octree(keys, 8, tree, counts)
        assert np.array_equal(t1, t2) and np.array_equal(c1, c2) and k1 == k2
        # perturb the keys so that merges occur as well
        keys = np.sort((keys.astype(np.uint64) // 3 * 2).astype(key_dtype(kb)))
        tree, counts = t1, oracle.node_counts(t1, keys)
        assert np.array_equal(counts, reference.node_counts(t1, keys))
    cs = np.array(sorted({0, 1, 0o30173, 0o3333333333, end_key(kb) - 1, end_key(kb)}), dtype=key_dtype(kb))
    assert np.array_equal(oracle.spanning_tree(cs), reference.spanning_tree(cs))


@pytest.mark.parametrize("kb,rb,hb", [(64, 64, 64), (64, 64, 32), (32, 32, 32), (64, 32, 32)])
@pytest.mark.parametrize("bc", [(0, 0, 0), (1, 1, 1)])
def test_halo_radii_rule_against_reference_halos_discover(oracle, kb, rb, hb, bc):
    """the halo search radius (max h of a leaf * 2 * ext, rounded to float) lives inline inside the reference's member
    function Halos::discover (R/halos/halos.hpp:168-180).  The reference's own discover + computeLayout run here for a
    pretended assignment [first, last) on one rank (oracle/_ref/libcstone_ref_domain.so): a leaf outside the assignment
    gets a non-empty layout range iff it was flagged.  oracle.halo_radii + oracle.find_halos must flag the same leaves."""
    import ctypes as C
    import os

    from oracle import oracle as orc

    path = os.path.join(os.path.dirname(orc.__file__), "_ref", "libcstone_ref_domain.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libcstone_ref_domain.so not built")
    lib = C.CDLL(path)
    box = Box([0, 1, 0, 2, -1, 1], bc)
    n = 30000
    x, y, z = random_cloud(n, box, rb, seed=kb + hb, kind="clustered")
    keys = oracle.compute_sfc_keys(HILBERT, kb, x, y, z, box)
    ks, order = oracle.sort_pairs(keys, np.arange(n))
    tree, counts = oracle.compute_octree(ks, 16)
    o = oracle.build_octree(tree)
    nl = tree.size - 1
    rng = np.random.default_rng(5)
    h_all = (0.02 * rng.uniform(0.2, 2.0, n)).astype(real_dtype(hb))
    layout_all = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    for first, last in ((0, nl // 3), (nl // 3, 2 * nl // 3), (nl // 2, nl)):
        for ext in (1.0, 1.3):
            h_loc = np.ascontiguousarray(h_all[layout_all[first]:layout_all[last]])
            lay = np.zeros(nl + 1, dtype=np.uint32)
            rc = lib.cstone_refdom_halo_discover(C.c_int(kb), C.c_int(rb), C.c_int(hb), p(o["prefixes"]), p(o["child_offsets"]),
                                                 p(o["internal_to_leaf"]), p(tree), p(counts), C.c_int(nl), C.c_int(first),
                                                 C.c_int(last), p(box.lim), p(box.bc), p(h_loc), C.c_float(ext), p(lay))
            assert rc == 0
            present = np.diff(lay.astype(np.int64)) > 0
            layout_loc = (layout_all[first:last + 1] - layout_all[first]).astype(np.uint32)
            radii = oracle.halo_radii(h_loc, layout_loc, first, last, nl, ext)
            flags = oracle.find_halos(HILBERT, o, tree, radii, box, first, last, rb)
            outside = np.ones(nl, dtype=bool)
            outside[first:last] = False
            nonempty = counts > 0
            assert np.array_equal(present[outside & nonempty], flags[outside & nonempty] != 0)
            assert present[first:last][counts[first:last] > 0].all()
            assert flags[outside].sum() > 0
