"""CPU tests: oracle (and, where built, the reference itself) against the literal known-answer vectors of the
reference's unit tests (tests/golden/reference_kats.json) and structural invariants."""
import json
import os

import numpy as np
import pytest

from helpers import Box, OctreeMaker, end_key, key_dtype, max_level, pad
from oracle.oracle import HILBERT, MORTON

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


@pytest.fixture(params=["oracle", "reference"])
def impl(request):
    return request.getfixturevalue(request.param)


def _tree(spec, kb):
    m = OctreeMaker(kb)
    for part in spec.split(".")[:]:
        inner = part[part.index("(") + 1:part.index(")")]
        m.divide(*[int(v) for v in inner.split(",") if v.strip()])
    return m.make()


@pytest.mark.parametrize("kb", [32, 64])
def test_morton_kats(impl, kb):
    k = KATS["morton_encode_level3"]
    shift = max_level(kb) - k["level"]
    ix, iy, iz = [v << shift for v in k["ixyz"]]
    assert impl.encode(MORTON, kb, ix, iy, iz) == pad(int(k["pad"][0], 2), k["pad"][1], kb)
    d = KATS["morton_decode32"]
    assert impl.decode(MORTON, 32, d["key"]) == tuple(d["xyz"])
    for c in KATS["morton_decode64"]["cases"]:
        out = impl.decode(MORTON, 64, int(c["key"], 16))
        if "xyz" in c:
            assert out == tuple(c["xyz"])
        else:
            assert out[2] == c["z"]


@pytest.mark.parametrize("kb", [32, 64])
def test_hilbert_first_order_and_ibox(impl, kb):
    h2m = KATS["hilbert_first_order"]["hilbert_to_morton"]
    half = (1 << max_level(kb)) // 2
    for xi in range(2):
        for yi in range(2):
            for zi in range(2):
                for off in (0, half - 1):
                    key = impl.encode(HILBERT, kb, half * xi + off, half * yi + off, half * zi + off)
                    octant = (key >> (3 * (max_level(kb) - 1))) & 7
                    assert h2m[octant] == 4 * xi + 2 * yi + zi
    k = KATS["hilbert_ibox"]
    start = pad(int(k["start_pad"][0], 8), k["start_pad"][1], kb)
    assert impl.node_ibox(HILBERT, kb, start, k["level"]) == tuple(v * half for v in k["box_in_half_cubes"])


@pytest.mark.parametrize("kb", [32, 64])
def test_hilbert_continuity_and_inversion(impl, kb):
    # T/sfc/hilbert.cpp:138-204
    for level in range(1, max_level(kb)):
        for octant in range(8 if level > 1 else 7):
            last = (octant + 1) * (1 << (3 * (max_level(kb) - level))) - 1
            a, b = impl.decode(HILBERT, kb, last), impl.decode(HILBERT, kb, last + 1)
            assert sum(abs(int(p) - int(q)) for p, q in zip(a, b)) == 1
    rng = np.random.default_rng(0)
    for _ in range(1000):
        p = tuple(int(v) for v in rng.integers(0, 1 << max_level(kb), 3))
        for curve in (MORTON, HILBERT):
            assert impl.decode(curve, kb, impl.encode(curve, kb, *p)) == p


@pytest.mark.parametrize("kb", [32, 64])
def test_span_sfc_range(impl, kb):
    for c in KATS["span_sfc_range"]["cases"]:
        a, b = int(c["a"], 8), int(c["b"], 8)
        # spanning tree of {0?, a, b, end}: the segment [a,b) must be tiled by exactly the listed nodes
        keys = sorted(set([0, a, b, end_key(kb)]))
        t = impl.spanning_tree(np.array(keys, dtype=key_dtype(kb)))
        seg = [int(v) for v in t if a <= int(v) < b]
        assert seg == [int(v, 8) for v in c["nodes"]]


@pytest.mark.parametrize("kb", [32, 64])
def test_node_counts_and_rebalance(impl, kb):
    k = KATS["node_counts"]
    tree = _tree(k["tree"], kb)
    codes = np.array(sorted(int(tree[i]) + d for i, d in k["codes_rel"]), dtype=key_dtype(kb))
    assert impl.node_counts(tree, codes).tolist() == k["counts"]

    k = KATS["rebalance_decision"]
    tree = _tree(k["tree"], kb)
    ops, conv = impl.node_ops(tree, np.array(k["counts"], dtype=np.uint32), k["bucket"])
    assert ops.tolist() == k["ops"] and conv == k["converged"]

    # single root stays (T/tree/csarray.cpp:131-150)
    root = OctreeMaker(kb).make()
    ops, conv = impl.node_ops(root, np.array([1], dtype=np.uint32), 4)
    assert ops.tolist() == [1] and conv

    # max-depth node cannot be split (T/tree/csarray.cpp:160-186)
    m = OctreeMaker(kb)
    for level in range(max_level(kb)):
        m.divide(*([0] * level))
    deep = m.make()
    counts = np.ones(deep.size - 1, dtype=np.uint32)
    counts[0] = 2
    ops, conv = impl.node_ops(deep, counts, 1)
    assert (ops == 1).all() and conv


@pytest.mark.parametrize("kb", [32, 64])
def test_update_octree_matches_rebalance_kat(impl, kb):
    """T/tree/csarray.cpp:188-206 through the public one-step update: craft keys that produce the KAT's op vector"""
    k = KATS["rebalance_tree"]
    tree = _tree(k["tree"], kb)
    ref = _tree(k["result"], kb)
    bucket = 4
    # node 0..7 (children of octant 0): empty -> merge; node 9 and 14: 5 keys -> split; others: 4 keys -> keep
    per_node = [0] * 8 + [4, 5, 4, 4, 4, 4, 5]
    keys = np.array(sorted(int(tree[i]) + j for i, c in enumerate(per_node) for j in range(c)), dtype=key_dtype(kb))
    counts = impl.node_counts(tree, keys)
    assert counts.tolist() == per_node
    ops, _ = impl.node_ops(tree, counts, bucket)
    assert ops.tolist() == k["ops"]
    new_tree, new_counts, conv = impl.update_octree(keys, bucket, tree, counts)
    assert not conv and np.array_equal(new_tree, ref)
    assert np.array_equal(new_counts, impl.node_counts(new_tree, keys))


def _check_connectivity(o, kb):
    """T/unit/tree/octree.cpp:43-90 restated on the flat arrays"""
    nn, ni = o["num_nodes"], o["num_internal"]
    pre, co, par = o["prefixes"], o["child_offsets"], o["parents"]
    assert np.all(np.diff(pre.astype(np.uint64)) > 0)
    def level(p):
        return (int(p).bit_length() - 1) // 3
    def start(p):
        nb = int(p).bit_length() - 1
        return (int(p) ^ (1 << nb)) << (3 * max_level(kb) - nb)
    for i in range(nn):
        if co[i]:
            c0 = co[i]
            for j in range(8):
                assert level(pre[c0 + j]) == level(pre[i]) + 1
                assert par[(c0 + j - 1) // 8] == i
            assert start(pre[c0]) == start(pre[i])
    assert (co[:nn] != 0).sum() == ni
    # leaf <-> internal maps are inverse permutations
    itl, lti = o["internal_to_leaf"], o["leaf_to_internal"]
    assert np.array_equal(itl[lti] + ni, np.arange(nn))


@pytest.mark.parametrize("kb", [32, 64])
def test_linked_octree_kats(impl, kb):
    # 4x4x4
    m = OctreeMaker(kb).divide()
    for i in range(8):
        m.divide(i)
    o = impl.build_octree(m.make())
    k = KATS["octree_4x4x4"]
    assert o["num_nodes"] == k["num_nodes"] == o["level_range"][-1]
    assert np.diff(o["level_range"])[:3].tolist() == k["level_counts"]
    _check_connectivity(o, kb)
    # irregular L3
    k = KATS["octree_irregular_l3"]
    m = OctreeMaker(kb)
    for d in k["divides"]:
        m.divide(*d)
    o = impl.build_octree(m.make())
    assert (o["num_nodes"], o["num_leaves"], o["num_internal"]) == (k["num_nodes"], k["num_leaves"], k["num_internal"])
    assert np.diff(o["level_range"])[:4].tolist() == k["level_counts"]
    _check_connectivity(o, kb)
    # root only
    o = impl.build_octree(OctreeMaker(kb).make())
    assert o["num_nodes"] == 1 and o["child_offsets"][0] == 0 and o["prefixes"][0] == 1
    # max-depth spanning tree
    cs = [int(v, 8) for v in KATS["octree_spanning"]["cornerstones_octal"]] + [end_key(kb) - 1, end_key(kb)]
    t = impl.spanning_tree(np.array(cs, dtype=key_dtype(kb)))
    _check_connectivity(impl.build_octree(t), kb)


@pytest.mark.parametrize("kb", [32, 64])
def test_find_halos_4x4x4(impl, kb):
    k = KATS["find_halos_4x4x4"]
    m = OctreeMaker(kb).divide()
    for i in range(8):
        m.divide(i)
    leaves = m.make()
    o = impl.build_octree(leaves)
    radii = np.full(64, k["radius"], dtype=np.float32)
    for first, last in k["ranges"]:
        flags = impl.find_halos(HILBERT, o, leaves, radii, Box([0, 1]), first, last)
        assert flags.sum() == k["num_flags"]
        assert flags[first:last].sum() == 0
