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
