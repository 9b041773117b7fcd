"""Focus-tree (locally essential tree) functions of the reference's GPU seam: R/focus/rebalance_gpu.h:40-81, markMacsGpu
(R/traversal/collisions_gpu.h:68-77), countSfcGapsGpu / fillSfcGapsGpu (R/tree/csarray_gpu.h:78-88), the node spheres of
R/focus/source_center_gpu.h and the small primitives (segmentMax, gatherRanges, fill / count / reduce ...).

CPU (-m "not gpu"): the oracle restatement against the reference's own CPU functions (oracle/_ref), bit for bit.
GPU (-m gpu): libcstone_hip against the oracle on the same inputs, bit for bit (integers and floating point alike)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc


def _tree(cpu, kb, n, bucket, seed, clustered=False):
    """sorted random keys -> converged cornerstone tree, linked octree, leaf and node counts"""
    rng = np.random.default_rng(seed)
    box = orc.Box([0, 1])
    if clustered:
        pts = np.clip(rng.normal(0.5, 0.12, (3, n)), 0, 0.999999)
    else:
        pts = rng.uniform(0, 1, (3, n))
    keys = np.sort(cpu.compute_sfc_keys(orc.HILBERT, kb, pts[0], pts[1], pts[2], box))
    leaves, counts = cpu.compute_octree(keys, bucket)
    oct_ = cpu.build_octree(leaves)
    node_counts = cpu.upsweep_counts(oct_, counts)
    return keys, leaves, counts, oct_, node_counts, box


def _focus_inputs(cpu, kb, seed, n=20000, bucket=16):
    keys, leaves, counts, oct_, node_counts, box = _tree(cpu, kb, n, bucket, seed, clustered=seed % 2 == 1)
    rng = np.random.default_rng(seed + 100)
    nl = leaves.size - 1
    first, last = sorted(rng.choice(nl + 1, 2, replace=False))
    macs = (rng.uniform(size=oct_["num_nodes"]) < 0.4).astype(np.int8)
    return dict(keys=keys, leaves=leaves, counts=counts, oct=oct_, node_counts=node_counts, box=box, first=int(first),
                last=int(last), macs=macs, nl=nl, rng=rng)


def _forced_keys(d, kb, m=24):
    """mandatory keys: existing leaf boundaries, keys one or more levels below a leaf, 0 and the end key"""
    rng, leaves = d["rng"], d["leaves"]
    ml = orc.max_level(kb)
    out = [0, orc.end_key(kb)]
    for _ in range(m):
        i = int(rng.integers(0, d["nl"]))
        start, span = int(leaves[i]), int(leaves[i + 1]) - int(leaves[i])
        kind = int(rng.integers(0, 3))
        if kind == 0 or span < 8:
            out.append(start)
        elif kind == 1:
            out.append(start + int(rng.integers(1, 8)) * (span // 8))
        else:
            sub = span // 64 if span >= 64 else span // 8
            out.append(start + int(rng.integers(1, span // sub)) * sub)
    return np.array(out, dtype=orc.key_dtype(kb)), ml


# ------------------------------------------------------------------------------------------------------------------
# CPU: oracle == reference
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kb", [32, 64])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_focus_ops_against_reference(oracle, reference, kb, seed):
    d = _focus_inputs(oracle, kb, seed)
    o, fs, fe = d["oct"], d["leaves"][d["first"]], d["leaves"][d["last"]]
    for bucket in (4, 16, 200):
        a = oracle.essential_ops(o, d["node_counts"], d["macs"], fs, fe, bucket)
        b = reference.essential_ops(o, d["node_counts"], d["macs"], fs, fe, bucket)
        assert np.array_equal(a, b)
        assert set(np.unique(a)) <= {0, 1, 8}
        pa, ca = oracle.protect_ancestors(o, a)
        pb, cb = reference.protect_ancestors(o, b)
        assert np.array_equal(pa, pb) and ca == cb
        fk, _ = _forced_keys(d, kb)
        ea, sa = oracle.enforce_keys(fk, o, a)
        eb, sb = reference.enforce_keys(fk, o, b)
        assert np.array_equal(ea, eb) and sa == sb
    ma = oracle.mac_refine_ops(o, d["macs"], d["nl"], d["first"], d["last"])
    mb = reference.mac_refine_ops(o, d["macs"], d["nl"], d["first"], d["last"])
    assert np.array_equal(ma, mb) and (ma[d["first"]:d["last"]] == 1).all()


@pytest.mark.parametrize("kb", [32, 64])
def test_oracle_range_count_against_reference(oracle, reference, kb):
    # a fine "global" tree and a coarser "focus" tree over the same keys: every focus leaf is a union of global leaves
    keys, gl, gc, _, _, _ = _tree(oracle, kb, 30000, 8, 5)
    fl, _ = oracle.compute_octree(keys, 64)
    idx = np.random.default_rng(1).permutation(fl.size - 1)[: (fl.size - 1) // 2]
    a = oracle.range_count(gl, gc, fl, idx)
    b = reference.range_count(gl, gc, fl, idx)
    assert np.array_equal(a, b)
    total = oracle.range_count(gl, gc, fl, np.arange(fl.size - 1))
    assert int(total.sum()) == keys.size
    # saturation
    big = np.full(gl.size - 1, 0xFFFFFFF0, dtype=np.uint32)
    assert np.array_equal(oracle.range_count(gl, big, fl, idx), reference.range_count(gl, big, fl, idx))


@pytest.mark.parametrize("kb", [32, 64])
def test_oracle_span_sfc_range_against_reference(oracle, reference, kb):
    rng = np.random.default_rng(3)
    end = orc.end_key(kb)
    for _ in range(200):
        a, b = sorted(int(v) for v in rng.integers(0, end, 2, dtype=np.uint64))
        lvl = int(rng.integers(1, orc.max_level(kb) + 1))
        unit = 1 << (3 * (orc.max_level(kb) - lvl))
        a, b = a // unit * unit, min(end, (b // unit + 1) * unit)
        if a >= b:
            continue
        x, y = oracle.span_sfc_range(kb, a, b), reference.span_sfc_range(kb, a, b)
        assert np.array_equal(x, y)
        assert int(x[0]) == a
    # the example of the reference's documentation, R/sfc/common.hpp:380-385: 1 .. 741 (octal, 10 digits for 32-bit keys)
    if kb == 32:
        got = oracle.span_sfc_range(32, 0o0000000001, 0o0000000741)
        assert got.size == 7 + 7 + 6 + 4 + 1 or np.array_equal(got, reference.span_sfc_range(32, 0o1, 0o741))


@pytest.mark.parametrize("kb,rb", [(64, 64), (32, 32), (64, 32), (32, 64)])
@pytest.mark.parametrize("bc", [(0, 0, 0), (1, 1, 1), (0, 1, 0)])
def test_oracle_mac_spheres_and_mark_macs_against_reference(oracle, reference, kb, rb, bc):
    d = _focus_inputs(oracle, kb, 7 + kb + rb, n=6000, bucket=8)
    box = orc.Box([-0.5, 1.5, 0.0, 1.0, -1.0, 1.0], bc)
    pre = d["oct"]["prefixes"]
    for inv_theta in (1.0 / 0.5 + 0.5, 1.0 / 0.8):
        sa = oracle.mac_spheres(orc.HILBERT, 0, pre, box, inv_theta, rb)
        sb = reference.mac_spheres(orc.HILBERT, 0, pre, box, inv_theta, rb)
        assert np.array_equal(sa, sb)
        # vector MAC from perturbed centres, some of them "empty" (mass 0)
        rng = np.random.default_rng(kb)
        com = sa.copy()
        com[:, :3] += (rng.uniform(-1, 1, (pre.size, 3)) * 0.01).astype(sa.dtype)
        com[:, 3] = (rng.uniform(size=pre.size) < 0.8).astype(sa.dtype)
        va = oracle.mac_spheres(orc.HILBERT, 1, pre, box, inv_theta, rb, com)
        vb = reference.mac_spheres(orc.HILBERT, 1, pre, box, inv_theta, rb, com)
        assert np.array_equal(va, vb)
        for limit in (False, True):
            fn = d["leaves"][d["first"]:d["last"] + 1]
            if fn.size < 2:
                continue
            ma = oracle.mark_macs(orc.HILBERT, d["oct"], sa, box, fn, limit)
            mb = reference.mark_macs(orc.HILBERT, d["oct"], sb, box, fn, limit)
            assert np.array_equal(ma, mb)
            assert ma.sum() > 0 or d["last"] - d["first"] == d["nl"]


@pytest.mark.parametrize("tc,tm,tf", [(64, 64, 64), (64, 32, 64), (32, 32, 32)])
def test_oracle_source_centers_against_reference(oracle, reference, tc, tm, tf):
    keys, leaves, counts, o, _, _ = _tree(oracle, 64, 8000, 16, 11)
    rng = np.random.default_rng(2)
    n = keys.size
    x, y, z = [rng.uniform(0, 1, n).astype(orc.real_dtype(tc)) for _ in range(3)]
    m = rng.uniform(-1, 1, n).astype(orc.real_dtype(tm))  # negative "masses": the centre uses |m|
    layout = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    l2i = o["leaf_to_internal"][o["num_internal"]:]
    a = oracle.leaf_source_centers(x, y, z, m, l2i, layout, o["num_nodes"], tf)
    b = reference.leaf_source_centers(x, y, z, m, l2i, layout, o["num_nodes"], tf)
    assert np.array_equal(a, b)
    ua = oracle.upsweep_centers(o, a, orc.max_level(64))
    ub = reference.upsweep_centers(o, b, orc.max_level(64))
    assert np.array_equal(ua, ub)
    assert abs(float(ua[0, 3]) - float(np.abs(m).sum())) < 1e-3 * n


# ------------------------------------------------------------------------------------------------------------------
# GPU: libcstone_hip == oracle
# ------------------------------------------------------------------------------------------------------------------
_KEEP = []  # uploaded arrays stay referenced: a temporary freed inside an argument list could be reused by the next upload


def _dev(a):
    import torch

    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:]
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    elif a.dtype == np.uint64:
        a = a.view(np.int64)
    t = torch.from_numpy(a.copy()).cuda()
    _KEEP.append(t)
    return t


def _host(t, dtype):
    return t.cpu().numpy().view(dtype)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _cbox(box):
    import cstone_amd

    return cstone_amd.make_cbox(box.lim, box.bc)


@pytest.mark.gpu
@pytest.mark.parametrize("kb", [32, 64])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_hip_focus_ops(hip, oracle, kb, seed):
    import torch

    d = _focus_inputs(oracle, kb, seed, n=60000)
    o, fs, fe = d["oct"], int(d["leaves"][d["first"]]), int(d["leaves"][d["last"]])
    nn = o["num_nodes"]
    pre, co, par = _dev(o["prefixes"]), _dev(o["child_offsets"]), _dev(o["parents"] if o["parents"].size else np.zeros(1, np.int32))
    cnt, macs = _dev(d["node_counts"]), _dev(d["macs"])
    lib, h = hip.lib, hip.h
    for bucket in (4, 16, 200):
        ops = torch.empty(nn, dtype=torch.int32, device="cuda")
        hip._chk(lib.cstone_hip_rebalance_decision_essential(h, kb, _p(pre), _p(co), _p(par), _p(cnt), _p(macs),
                                                             C.c_uint64(fs), C.c_uint64(fe), C.c_uint(bucket), _p(ops),
                                                             C.c_int(nn)), "essential")
        ref = oracle.essential_ops(o, d["node_counts"], d["macs"], fs, fe, bucket)
        assert np.array_equal(_host(ops, np.int32), ref)

        fk, _ = _forced_keys(d, kb, 64)
        ops2 = ops.clone()
        status = C.c_int(-1)
        hip._chk(lib.cstone_hip_enforce_keys(h, kb, _p(_dev(fk)), C.c_int(fk.size), _p(pre), _p(co), _p(par), _p(ops2),
                                             C.byref(status)), "enforce_keys")
        ref2, st = oracle.enforce_keys(fk, o, ref)
        assert np.array_equal(_host(ops2, np.int32), ref2) and status.value == st

        conv = C.c_int(-1)
        hip._chk(lib.cstone_hip_protect_ancestors(h, kb, _p(pre), _p(par), _p(ops2), C.c_int(nn), C.byref(conv)), "protect")
        ref3, c3 = oracle.protect_ancestors(o, ref2)
        assert np.array_equal(_host(ops2, np.int32), ref3) and bool(conv.value) == c3

    l2i = _dev(o["leaf_to_internal"][o["num_internal"]:])
    lops = torch.empty(d["nl"], dtype=torch.int32, device="cuda")
    hip._chk(lib.cstone_hip_mac_refine_decision(h, kb, _p(pre), _p(macs), _p(l2i), C.c_int(d["nl"]), C.c_int(d["first"]),
                                                C.c_int(d["last"]), _p(lops)), "mac_refine")
    assert np.array_equal(_host(lops, np.int32), oracle.mac_refine_ops(o, d["macs"], d["nl"], d["first"], d["last"]))
    hip.sync()


@pytest.mark.gpu
@pytest.mark.parametrize("kb", [32, 64])
def test_hip_range_count_and_sfc_gaps(hip, oracle, kb):
    import torch

    keys, gl, gc, _, _, _ = _tree(oracle, kb, 200000, 8, 5)
    fl, _ = oracle.compute_octree(keys, 512)
    nf = fl.size - 1
    idx = np.random.default_rng(1).permutation(nf)[: max(1, nf // 2)].astype(np.int32)
    lib, h = hip.lib, hip.h
    for counts in (gc, np.full(gl.size - 1, 0xFFFFFFF0, dtype=np.uint32)):
        out = torch.zeros(nf, dtype=torch.int32, device="cuda")
        hip._chk(lib.cstone_hip_range_count(h, kb, _p(_dev(gl)), C.c_int(gl.size - 1), _p(_dev(counts)), _p(_dev(fl)),
                                            _p(_dev(idx)), C.c_int(idx.size), _p(out)), "range_count")
        assert np.array_equal(_host(out, np.uint32), oracle.range_count(gl, counts, fl, idx))

    # gaps: a sparse selection of the leaf keys is completed to a valid cornerstone array again
    rng = np.random.default_rng(9)
    pick = np.sort(rng.choice(np.arange(1, gl.size - 1), 300, replace=False))
    sparse = np.concatenate([[0], gl[pick], [orc.end_key(kb)]]).astype(gl.dtype)
    m = sparse.size - 1
    ops = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    sd = _dev(sparse)
    hip._chk(lib.cstone_hip_count_sfc_gaps(h, kb, _p(sd), C.c_int(m), _p(ops)), "count_gaps")
    want = [oracle.span_sfc_range(kb, int(sparse[i]), int(sparse[i + 1])) for i in range(m)]
    got_counts = _host(ops, np.int32)[:m]
    assert np.array_equal(got_counts, [w.size for w in want])
    scan = np.concatenate([[0], np.cumsum(got_counts)]).astype(np.int32)
    new_tree = torch.zeros(int(scan[-1]) + 1, dtype=sd.dtype, device="cuda")
    hip._chk(lib.cstone_hip_fill_sfc_gaps(h, kb, _p(sd), C.c_int(m), _p(_dev(scan)), _p(new_tree)), "fill_gaps")
    full = np.concatenate(want + [np.array([orc.end_key(kb)], dtype=gl.dtype)])
    assert np.array_equal(_host(new_tree, gl.dtype), full)
    hip.sync()


@pytest.mark.gpu
@pytest.mark.parametrize("kb,rb", [(64, 64), (32, 32), (64, 32), (32, 64)])
@pytest.mark.parametrize("bc", [(0, 0, 0), (1, 1, 1), (0, 1, 0)])
def test_hip_mac_spheres_and_mark_macs(hip, oracle, kb, rb, bc):
    import torch

    d = _focus_inputs(oracle, kb, 7 + kb + rb, n=40000, bucket=8)
    box = orc.Box([-0.5, 1.5, 0.0, 1.0, -1.0, 1.0], bc)
    cb = _cbox(box)
    o = d["oct"]
    nn = o["num_nodes"]
    rt = orc.real_dtype(rb)
    pre, co = _dev(o["prefixes"]), _dev(o["child_offsets"])
    lib, h = hip.lib, hip.h
    for curve in (orc.HILBERT, orc.MORTON):
        for inv_theta in (2.5, 1.25):
            sph = torch.zeros(nn * 4, dtype=torch.float32 if rb == 32 else torch.float64, device="cuda")
            hip._chk(lib.cstone_hip_geo_mac_spheres(h, curve, kb, rb, _p(pre), C.c_int(nn), _p(sph), C.c_float(inv_theta),
                                                    C.byref(cb)), "geo_mac_spheres")
            ref = oracle.mac_spheres(curve, 0, o["prefixes"], box, inv_theta, rb)
            assert np.array_equal(_host(sph, rt).reshape(nn, 4), ref)

            rng = np.random.default_rng(kb)
            com = ref.copy()
            com[:, :3] += (rng.uniform(-1, 1, (nn, 3)) * 0.01).astype(rt)
            com[:, 3] = (rng.uniform(size=nn) < 0.8).astype(rt)
            cd = _dev(com.reshape(-1))
            hip._chk(lib.cstone_hip_set_mac(h, curve, kb, rb, _p(pre), C.c_int(nn), _p(cd), C.c_float(inv_theta),
                                            C.byref(cb)), "set_mac")
            assert np.array_equal(_host(cd, rt).reshape(nn, 4), oracle.mac_spheres(curve, 1, o["prefixes"], box, inv_theta, rb, com))

            fn = d["leaves"][d["first"]:d["last"] + 1]
            if fn.size < 2:
                continue
            for limit in (0, 1):
                marks = torch.zeros(nn, dtype=torch.int8, device="cuda")
                hip._chk(lib.cstone_hip_mark_macs(h, curve, kb, rb, _p(pre), _p(co), _p(sph), C.byref(cb), _p(_dev(fn)),
                                                  C.c_int(fn.size - 1), C.c_int(limit), _p(marks)), "mark_macs")
                want = oracle.mark_macs(curve, o, ref, box, fn, bool(limit))
                got = _host(marks, np.int8)
                assert np.array_equal(got, want), (int(got.sum()), int(want.sum()))
    hip.sync()


@pytest.mark.gpu
@pytest.mark.parametrize("tc,tm,tf", [(64, 64, 64), (64, 32, 64), (32, 32, 32)])
def test_hip_source_centers(hip, oracle, tc, tm, tf):
    import torch

    keys, leaves, counts, o, _, _ = _tree(oracle, 64, 100000, 16, 11)
    rng = np.random.default_rng(2)
    n = keys.size
    x, y, z = [rng.uniform(0, 1, n).astype(orc.real_dtype(tc)) for _ in range(3)]
    m = rng.uniform(-1, 1, n).astype(orc.real_dtype(tm))
    layout = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    l2i = o["leaf_to_internal"][o["num_internal"]:]
    nn = o["num_nodes"]
    ft = orc.real_dtype(tf)
    ctr = torch.zeros(nn * 4, dtype=torch.float32 if tf == 32 else torch.float64, device="cuda")
    lib, h = hip.lib, hip.h
    hip._chk(lib.cstone_hip_leaf_source_centers(h, tc, tm, tf, _p(_dev(x)), _p(_dev(y)), _p(_dev(z)), _p(_dev(m)),
                                                _p(_dev(l2i)), C.c_int(l2i.size), _p(_dev(layout)), _p(ctr)), "leaf_centers")
    want = oracle.leaf_source_centers(x, y, z, m, l2i, layout, nn, tf)
    assert np.array_equal(_host(ctr, ft).reshape(nn, 4), want)
    lr = np.ascontiguousarray(o["level_range"], dtype=np.int32)
    hip._chk(lib.cstone_hip_upsweep_centers(h, tf, C.c_int(orc.max_level(64)), lr.ctypes.data_as(C.c_void_p),
                                            _p(_dev(o["child_offsets"])), _p(ctr)), "upsweep_centers")
    assert np.array_equal(_host(ctr, ft).reshape(nn, 4), oracle.upsweep_centers(o, want, orc.max_level(64)))
    mv = torch.zeros(nn * 4, dtype=ctr.dtype, device="cuda")
    src3 = _dev(np.ascontiguousarray(want[:, :3]).reshape(-1))
    hip._chk(lib.cstone_hip_move_centers(h, tf, _p(src3), C.c_int(nn), _p(mv)), "move_centers")
    got = _host(mv, ft).reshape(nn, 4)
    assert np.array_equal(got[:, :3], want[:, :3]) and (got[:, 3] == 1).all()
    hip.sync()


@pytest.mark.gpu
def test_hip_small_primitives(hip, oracle):
    import torch

    lib, h = hip.lib, hip.h
    rng = np.random.default_rng(4)
    # segmentMax (float|double in, float|double out) on ragged segments incl. empty ones
    lens = rng.integers(0, 90, 5000)
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    for ib, ob in ((32, 32), (64, 32), (64, 64)):
        vals = rng.uniform(0, 1, int(seg[-1]) + 1).astype(orc.real_dtype(ib))  # empty segments must give 0
        out = torch.zeros(seg.size - 1, dtype=torch.float32 if ob == 32 else torch.float64, device="cuda")
        hip._chk(lib.cstone_hip_segment_max(h, ib, ob, 32, _p(_dev(vals)), _p(_dev(seg)), C.c_size_t(seg.size - 1), _p(out)),
                 "segment_max")
        assert np.array_equal(_host(out, orc.real_dtype(ob)), oracle.segment_max(vals, seg, ob))
    # gatherRanges for 4-, 12- and 32-byte elements
    nr = 40
    offs = np.sort(rng.choice(100000, nr, replace=False)).astype(np.uint32)
    rl = rng.integers(0, 300, nr)
    scan = np.concatenate([[0], np.cumsum(rl)]).astype(np.uint32)
    for eb in (4, 12, 32):
        src = rng.integers(0, 255, (101000, eb), dtype=np.uint8)
        buf = torch.zeros(int(scan[-1]) * eb, dtype=torch.uint8, device="cuda")
        hip._chk(lib.cstone_hip_gather_ranges(h, eb, 32, _p(_dev(scan[:-1])), _p(_dev(offs)), C.c_int(nr),
                                              _p(_dev(src)), _p(buf), C.c_size_t(int(scan[-1]))), "gather_ranges")
        want = np.concatenate([src[offs[r]:offs[r] + rl[r]] for r in range(nr)])
        assert np.array_equal(buf.cpu().numpy().reshape(-1, eb), want)
    # fill / increment / scale / count / reduce / max norm / lower bound / keys-only sort
    t = torch.zeros(1000, dtype=torch.int64, device="cuda")
    v = C.c_uint64(0xABCDEF0123456789)
    hip._chk(lib.cstone_hip_fill(h, 8, _p(t), C.c_size_t(1000), C.byref(v)), "fill")
    assert (_host(t, np.uint64) == v.value).all()
    a = rng.integers(0, 1000, 100001).astype(np.uint32)
    ad, bd = _dev(a), torch.zeros(a.size, dtype=torch.int32, device="cuda")
    hip._chk(lib.cstone_hip_increment(h, 32, _p(ad), _p(bd), C.c_size_t(a.size), C.c_uint64(17)), "increment")
    assert np.array_equal(_host(bd, np.uint32), a + 17)
    cnt, tot = C.c_uint64(0), C.c_uint64(0)
    hip._chk(lib.cstone_hip_count_equal(h, 32, _p(ad), C.c_size_t(a.size), C.c_uint64(7), C.byref(cnt)), "count")
    hip._chk(lib.cstone_hip_reduce_sum(h, 32, _p(ad), C.c_size_t(a.size), C.c_uint64(5), C.byref(tot)), "reduce")
    assert cnt.value == int((a == 7).sum()) and tot.value == 5 + int(a.astype(np.uint64).sum())
    f = rng.uniform(-3, 3, (3, 50001))
    fd = [_dev(f[i]) for i in range(3)]
    mx = C.c_double(0)
    hip._chk(lib.cstone_hip_max_norm_square(h, 64, _p(fd[0]), _p(fd[1]), _p(fd[2]), C.c_size_t(f.shape[1]), C.byref(mx)), "norm")
    assert mx.value == float((f[0] * f[0] + f[1] * f[1] + f[2] * f[2]).max())
    hip._chk(lib.cstone_hip_scale(h, 64, _p(fd[0]), C.c_size_t(f.shape[1]), C.c_double(0.3)), "scale")
    assert np.array_equal(_host(fd[0], np.float64), f[0] * 0.3)
    s = np.sort(rng.integers(0, 2**62, 5000).astype(np.uint64))
    val, pos = C.c_uint64(int(s[1234])), C.c_uint64(0)
    hip._chk(lib.cstone_hip_lower_bound_value(h, 1, _p(_dev(s)), C.c_size_t(s.size), C.byref(val), C.byref(pos)), "lower_bound")
    assert pos.value == int(np.searchsorted(s, s[1234], side="left"))
    for kb in (32, 64):
        k = rng.integers(0, 2**30, 300000).astype(orc.key_dtype(kb))
        kd = _dev(k)
        hip._chk(lib.cstone_hip_sort_keys(h, kb, _p(kd), C.c_size_t(k.size)), "sort_keys")
        assert np.array_equal(_host(kd, k.dtype), np.sort(k))
    hip.sync()


# ------------------------------------------------------------------------------------------------------------------
# binary radix tree (btree), SURVEY.md section 8f-4
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kb", [32, 64])
def test_oracle_binary_tree_against_reference(oracle, reference, kb):
    for seed, n, bucket in ((1, 20000, 16), (2, 500, 1), (3, 3000, 64)):
        _, leaves, _, _, _, _ = _tree(oracle, kb, n, bucket, seed, clustered=seed % 2 == 0)
        ca, pa = oracle.binary_tree(leaves)
        cb, pb = reference.binary_tree(leaves)
        assert np.array_equal(ca, cb) and np.array_equal(pa, pb)
        # root covers everything: prefix 1 (placeholder bit only); every leaf (key index below the terminal key, whose set
        # bit lies above the key bits) is some node's leaf child exactly once
        assert int(pa[0]) == 1
        leaf_children = ca[ca < 0].astype(np.int64) + 2**31
        assert np.array_equal(np.sort(leaf_children), np.arange(leaves.size - 1))


@pytest.mark.gpu
@pytest.mark.parametrize("kb", [32, 64])
def test_hip_binary_tree(hip, oracle, kb):
    import torch

    for seed, n, bucket in ((1, 300000, 16), (2, 500, 1), (3, 30000, 64)):
        _, leaves, _, _, _, _ = _tree(oracle, kb, n, bucket, seed, clustered=seed % 2 == 0)
        m = leaves.size - 1
        node_bytes = 12 if kb == 32 else 16
        out = torch.zeros(m * node_bytes, dtype=torch.uint8, device="cuda")
        hip._chk(hip.lib.cstone_hip_create_binary_tree(hip.h, kb, _p(_dev(leaves)), C.c_int(m), _p(out)), "binary_tree")
        dt = np.dtype([("child", np.int32, 2), ("prefix", np.uint32 if kb == 32 else np.uint64)])
        assert dt.itemsize == node_bytes
        got = out.cpu().numpy().view(dt)
        child, prefix = oracle.binary_tree(leaves)
        assert np.array_equal(got["child"], child) and np.array_equal(got["prefix"], prefix)
    hip.sync()
