"""Target particle groups (computeFixedGroups / computeGroupSplits, R/traversal/groups_gpu.cu:41-151).

The reference implements them for the GPU only, so its own code cannot be run here; the oracle's restatement is pinned
by the literal known answers of the reference's unit test (R/../test/unit_cuda/traversal/groups.cu:55-262: fixed groups,
findSplits, the makeSplits table, groupVolumes incl. the knife-edge tolerance 1.01 / 0.99) -- the CPU tests below --
and the HIP kernels are compared with the oracle, and run through the same known answers, in the GPU tests."""
import numpy as np
import pytest

from helpers import Box, OctreeMaker, sorted_cloud
from oracle.oracle import HILBERT, MORTON


# ---------------------------------------------------------------------------------------------- known answers
def _group_volumes_case():
    """TEST(TargetGroups, groupVolumes), groups.cu:197-262"""
    first, last = 4, 128
    box = Box([0, last] * 3)
    leaves = OctreeMaker(64).divide().divide(2).make()
    counts = np.array([4, 1, 8, 8, 8, 8, 31, 8, 8, 8, 16, 16, 16, 0, 0], dtype=np.uint32)
    layout = np.zeros(counts.size + 1, dtype=np.uint32)
    layout[1:] = np.cumsum(counts)
    x = np.arange(last, dtype=np.float64)
    x[5] -= 0.01
    dist_crit = np.cbrt(128.0 * 128.0 * 128.0 / 64)
    return first, last, box, leaves, layout, x, dist_crit


def _tol(dist_crit, f):
    # `float tolFactor = std::sqrt(3.0) / distCrit * 1.01` evaluated in double, then narrowed
    return float(np.float32(np.sqrt(3.0) / dist_crit * f))


def test_fixed_groups_kat(oracle):
    assert oracle.fixed_groups(4, 34, 8).tolist() == [4, 12, 20, 28, 34]
    assert oracle.fixed_groups(0, 64, 64).tolist() == [0, 64]
    assert oracle.fixed_groups(7, 7, 64).tolist() == [7]


def test_find_splits_kat(oracle):
    """TEST(TargetGroups, findSplits), groups.cu:55-113: splits behind positions 0, 31 and 33"""
    t = np.arange(64, dtype=np.float64)
    pos = np.stack([t, t, t], axis=1)
    pos[0] = -1
    pos[31] -= 0.5
    pos[33] -= 0.5
    bits = int(oracle.find_splits(pos, 3.01)[0])
    assert bin(bits).count("1") == 3 and bits == (1 << 0) | (1 << 31) | (1 << 33)
    # two words: the last lane of word 0 looks at the first lane of word 1, the last lane of the run at itself
    t = np.arange(128, dtype=np.float64)
    pos = np.stack([t, t, t], axis=1)
    pos[64:] += 5
    w = oracle.find_splits(pos, 3.01)
    assert int(w[0]) == 1 << 63 and int(w[1]) == 0


@pytest.mark.parametrize("width", [32, 64])
def test_make_splits_kat(oracle, width):
    """TEST(TargetGroups, makeSplits), groups.cu:117-195, as two 32-bit words and as one 64-bit word"""

    def run(a, b):
        masks = [a, b] if width == 32 else [(b << 32) + a]
        return oracle.make_splits(masks, width).tolist()

    assert run(0, 0) == [64]
    assert run(1, 0) == [1, 63]
    assert run(0, 1 << 30) == [63, 1]
    assert run(2, 0) == [2, 62]
    assert run(3, 0) == [1, 1, 62]
    assert run(1 << 31, 1) == [32, 1, 31]
    assert run(0, 8) == [36, 28]
    got = run(0xFFFFFFFF, 0x6FFFFFFF)
    assert got[:63] == [1] * 60 + [2] + [1] * 2
    assert run(0xFFFFFFFF, 0x7FFFFFFF)[:63] == [1] * 63


def test_group_volumes_kat(oracle):
    first, last, box, leaves, layout, x, dist_crit = _group_volumes_case()
    g = oracle.group_splits(first, last, x, x, x, leaves, layout, box, 64, _tol(dist_crit, 1.01))
    assert g.tolist() == [4, 6, 68, 128]
    # tolerance 0.99: every pair splits -> 64 and 60 groups of one particle
    g = oracle.group_splits(first, last, x, x, x, leaves, layout, box, 64, _tol(dist_crit, 0.99))
    assert g.tolist() == list(range(4, 129))


def _cloud(oracle, kb, rb, n, seed, kind, curve=HILBERT, bucket=16):
    box = Box([-1, 1] * 3, (0, 1, 0))
    x, y, z, keys = sorted_cloud(oracle, curve, kb, n, box, rb, seed, kind)
    h = (0.06 * np.random.default_rng(seed).uniform(0.6, 1.4, n)).astype(x.dtype)
    tree, counts = oracle.compute_octree(keys, bucket)
    layout = np.zeros(counts.size + 1, dtype=np.uint32)
    layout[1:] = np.cumsum(counts)
    return box, keys, x, y, z, h, tree, layout


@pytest.mark.parametrize("group_size", [64, 128])
def test_group_splits_properties(oracle, group_size):
    """groups tile [first,last), none is longer than group_size, and a tighter tolerance only adds boundaries"""
    box, keys, x, y, z, h, tree, layout = _cloud(oracle, 64, 64, 5000, 3, "clustered")
    first, last = 13, 4990
    prev = None
    for tol in (4.0, 1.5, 0.5):
        g = oracle.group_splits(first, last, x, y, z, tree, layout, box, group_size, tol).astype(np.int64)
        assert g[0] == first and g[-1] == last and np.all(np.diff(g) > 0) and np.diff(g).max() <= group_size
        fixed = np.arange(first, last, group_size)
        assert np.isin(fixed, g).all()
        if prev is not None:
            assert np.isin(prev, g).all() and g.size > prev.size
        prev = g


# ---------------------------------------------------------------------------------------------- GPU parity
def _dev(a):
    import torch

    if a.dtype == np.uint32:
        a = a.view(np.int32)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_hip_fixed_groups(hip, oracle):
    for first, last, gs in [(4, 34, 8), (0, 64, 64), (7, 7, 64), (3, 100003, 64), (0, 1, 128)]:
        got = hip.compute_fixed_groups(first, last, gs).cpu().numpy().view(np.uint32)
        assert got.tolist() == oracle.fixed_groups(first, last, gs).tolist()


@pytest.mark.gpu
def test_hip_group_volumes_kat(hip):
    from cstone_amd import make_cbox

    first, last, box, leaves, layout, x, dist_crit = _group_volumes_case()
    cb = make_cbox(box.lim, box.bc)
    xd = _dev(x)
    g = hip.compute_group_splits(first, last, xd, xd, xd, _dev(leaves), _dev(layout), cb, 64, _tol(dist_crit, 1.01))
    assert g.cpu().tolist() == [4, 6, 68, 128]
    g = hip.compute_group_splits(first, last, xd, xd, xd, _dev(leaves), _dev(layout), cb, 64, _tol(dist_crit, 0.99))
    assert g.cpu().tolist() == list(range(4, 129))


@pytest.mark.gpu
@pytest.mark.parametrize("kb,rb", [(64, 64), (64, 32), (32, 32), (32, 64)])
@pytest.mark.parametrize("group_size", [64, 128])
def test_hip_group_splits_match_oracle(hip, oracle, kb, rb, group_size):
    from cstone_amd import make_cbox

    for kind, n, seed in [("uniform", 20000, 1), ("clustered", 30011, 2)]:
        box, keys, x, y, z, h, tree, layout = _cloud(oracle, kb, rb, n, seed, kind, curve=MORTON if seed == 1 else HILBERT)
        cb = make_cbox(box.lim, box.bc)
        xd, yd, zd, td, ld = _dev(x), _dev(y), _dev(z), _dev(tree), _dev(layout)
        for first, last, tol in [(0, n, 2.0), (17, n - 5, 0.7), (100, 163, 1.0), (5, 6, 1.0), (0, n, 1e-3)]:
            want = oracle.group_splits(first, last, x, y, z, tree, layout, box, group_size, tol)
            got = hip.compute_group_splits(first, last, xd, yd, zd, td, ld, cb, group_size, tol)
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want), (kind, first, last, tol)


@pytest.mark.gpu
def test_hip_group_splits_errors(hip):
    from cstone_amd import CstoneError, make_cbox

    first, last, box, leaves, layout, x, dist_crit = _group_volumes_case()
    cb = make_cbox(box.lim, box.bc)
    xd, ld, yd = _dev(x), _dev(leaves), _dev(layout)
    with pytest.raises(CstoneError):  # the reference throws "Unsupported spatial group size"
        hip.compute_group_splits(first, last, xd, xd, xd, ld, yd, cb, 32, 1.0)
    with pytest.raises(CstoneError, match="capacity"):
        hip.compute_group_splits(first, last, xd, xd, xd, ld, yd, cb, 64, _tol(dist_crit, 0.99), capacity=10)
    g = hip.compute_group_splits(9, 9, xd, xd, xd, ld, yd, cb, 64, 1.0)
    assert g.cpu().tolist() == [9]


@pytest.mark.gpu
@pytest.mark.parametrize("rb", [32, 64])
def test_hip_find_neighbors_over_groups_equals_fixed_targets(hip, oracle, rb):
    """the neighbor lists do not depend on how the targets are grouped: split groups, fixed groups of 64 and of 100
    particles (walked 64 at a time) all give the rows of find_neighbors, which the oracle pins"""
    import torch
    from cstone_amd import make_cbox

    n, ngmax = 20000, 150
    box, keys, x, y, z, h, tree, layout = _cloud(oracle, 64, rb, n, 5, "clustered", bucket=64)
    cb = make_cbox(box.lim, box.bc)
    oct_o = oracle.build_octree(tree)
    centers_o, sizes_o = oracle.node_centers(HILBERT, oct_o["prefixes"], box, rb)
    first, last = 100, n - 77
    want_idx, want_nc = oracle.find_neighbors(x, y, z, h, first, last, box, oct_o, layout, centers_o, sizes_o, ngmax)

    xd, yd, zd, hd, td, ld = _dev(x), _dev(y), _dev(z), _dev(h), _dev(tree), _dev(layout)
    octd = {k: _dev(v) for k, v in oct_o.items() if isinstance(v, np.ndarray)}
    cd, sd = _dev(centers_o), _dev(sizes_o)
    for groups in (hip.compute_group_splits(first, last, xd, yd, zd, td, ld, cb, 64, 1.0),
                   hip.compute_fixed_groups(first, last, 64), hip.compute_fixed_groups(first, last, 100)):
        nidx, nc = hip.find_neighbors_groups(xd, yd, zd, hd, first, last, groups, cb, octd, ld, cd, sd, ngmax)
        torch.cuda.synchronize()
        nc = nc.cpu().numpy().view(np.uint32)
        assert np.array_equal(nc, want_nc)
        nidx = nidx.cpu().numpy().view(np.uint32)
        for i in range(0, last - first, 37):
            k = min(int(nc[i]), ngmax)
            assert np.array_equal(nidx[i, :k], want_idx[i, :k])
