"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

ctypes front-end for the two CPU checkers:

* ``Oracle()``     -> oracle/libcstone_oracle.so   (our restatement, cstone_oracle.hpp)
* ``Reference()``  -> oracle/_ref/libcstone_ref.so (the reference's own headers, built by oracle/Makefile)

Both expose the same numpy-level methods so tests can run one input through oracle, reference and
the HIP library and compare.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

MORTON, HILBERT = 0, 1

_c_p = C.c_void_p


def key_dtype(key_bits):
    return {32: np.uint32, 64: np.uint64}[key_bits]


def real_dtype(real_bits):
    return {32: np.float32, 64: np.float64}[real_bits]


def max_level(key_bits):
    return {32: 10, 64: 21}[key_bits]


def end_key(key_bits):
    return 1 << (3 * max_level(key_bits))


def _p(a):
    return a.ctypes.data_as(_c_p)


def build(target="all"):
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)


class Box:
    """xmin,xmax,ymin,ymax,zmin,zmax + boundary type per axis (0 open, 1 periodic, 2 fixed)."""

    def __init__(self, lim, bc=(0, 0, 0)):
        if np.isscalar(lim[0]) and len(lim) == 2:
            lim = [lim[0], lim[1]] * 3
        self.lim = np.ascontiguousarray(lim, dtype=np.float64)
        self.bc = np.ascontiguousarray(bc, dtype=np.int32)
        assert self.lim.shape == (6,) and self.bc.shape == (3,)


class _CpuImpl:
    prefix = None
    libpath = None

    def __init__(self):
        if not os.path.exists(self.libpath):
            raise FileNotFoundError(self.libpath)
        self.lib = C.CDLL(self.libpath)
        f = self._f("encode")
        f.restype = C.c_uint64
        self._f("decode").restype = None
        self._f("node_ibox").restype = None

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def has(self, name):
        return hasattr(self.lib, self.prefix + name)

    # ---- scalar helpers
    def encode(self, curve, key_bits, ix, iy, iz):
        return int(self._f("encode")(C.c_int(curve), C.c_int(key_bits), C.c_uint(ix), C.c_uint(iy), C.c_uint(iz)))

    def decode(self, curve, key_bits, key):
        out = (C.c_uint * 3)()
        self._f("decode")(C.c_int(curve), C.c_int(key_bits), C.c_uint64(key), out)
        return tuple(out)

    def node_ibox(self, curve, key_bits, key, level):
        out = (C.c_int * 6)()
        self._f("node_ibox")(C.c_int(curve), C.c_int(key_bits), C.c_uint64(key), C.c_uint(level), out)
        return tuple(out)

    # ---- arrays
    def compute_sfc_keys(self, curve, key_bits, x, y, z, box, keys=None):
        real_bits = x.dtype.itemsize * 8
        n = x.size
        if keys is None:
            keys = np.zeros(n, dtype=key_dtype(key_bits))
        rc = self._f("compute_sfc_keys")(C.c_int(curve), C.c_int(key_bits), C.c_int(real_bits), _p(x), _p(y), _p(z),
                                         _p(keys), C.c_size_t(n), _p(box.lim), _p(box.bc))
        assert rc == 0, rc
        return keys

    def sort_pairs(self, keys, vals):
        """stable; returns (sorted keys, permuted vals) as new arrays"""
        k = keys.copy()
        v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        rc = self._f("sort_pairs")(C.c_int(k.dtype.itemsize * 8), _p(k), _p(v), C.c_size_t(k.size))
        assert rc == 0, rc
        return k, v

    def node_counts(self, tree, keys, max_count=0xFFFFFFFF):
        nn = tree.size - 1
        counts = np.zeros(nn, dtype=np.uint32)
        rc = self._f("node_counts")(C.c_int(tree.dtype.itemsize * 8), _p(tree), _p(counts), C.c_int(nn), _p(keys),
                                    C.c_size_t(keys.size), C.c_uint(max_count))
        assert rc == 0, rc
        return counts

    def node_ops(self, tree, counts, bucket):
        nn = tree.size - 1
        ops = np.zeros(nn + 1, dtype=np.int32)
        conv = C.c_int(0)
        rc = self._f("node_ops")(C.c_int(tree.dtype.itemsize * 8), _p(tree), C.c_int(nn), _p(counts), C.c_uint(bucket),
                                 _p(ops), C.byref(conv))
        assert rc == 0, rc
        return ops[:nn], bool(conv.value)

    def update_octree(self, keys, bucket, tree, counts, max_count=0xFFFFFFFF):
        """one rebalance step; returns (tree, counts, converged)"""
        kb = keys.dtype.itemsize * 8
        nl = tree.size - 1
        cap = max(8 * nl, 4096 * nl if nl < 64 else 8 * nl) + 8
        while True:
            t = np.zeros(cap + 1, dtype=keys.dtype)
            c = np.zeros(cap, dtype=np.uint32)
            t[:nl + 1] = tree
            c[:nl] = counts
            num = C.c_int(nl)
            conv = C.c_int(0)
            rc = self._f("update_octree")(C.c_int(kb), _p(keys), C.c_size_t(keys.size), C.c_uint(bucket), _p(t),
                                          _p(c), C.byref(num), C.c_int(cap), C.c_uint(max_count), C.byref(conv))
            if rc == -2:
                cap *= 8
                continue
            assert rc == 0, rc
            return t[:num.value + 1].copy(), c[:num.value].copy(), bool(conv.value)

    def compute_octree(self, keys, bucket, max_count=0xFFFFFFFF):
        kb = keys.dtype.itemsize * 8
        cap = max(64, 4 * keys.size // max(1, bucket) + 4096)
        while True:
            t = np.zeros(cap + 1, dtype=keys.dtype)
            c = np.zeros(cap, dtype=np.uint32)
            num = C.c_int(0)
            iters = C.c_int(0)
            rc = self._f("compute_octree")(C.c_int(kb), _p(keys), C.c_size_t(keys.size), C.c_uint(bucket), _p(t),
                                           _p(c), C.byref(num), C.c_int(cap), C.c_uint(max_count), C.byref(iters))
            if rc == -2:
                cap = num.value + 1
                continue
            assert rc == 0, rc
            return t[:num.value + 1].copy(), c[:num.value].copy()

    def spanning_tree(self, span_keys):
        kb = span_keys.dtype.itemsize * 8
        cap = 64 * span_keys.size * max_level(kb)
        t = np.zeros(cap + 1, dtype=span_keys.dtype)
        num = C.c_int(0)
        rc = self._f("spanning_tree")(C.c_int(kb), _p(span_keys), C.c_int(span_keys.size), _p(t), C.c_int(cap),
                                      C.byref(num))
        assert rc == 0, rc
        return t[:num.value + 1].copy()

    def build_octree(self, leaves):
        kb = leaves.dtype.itemsize * 8
        nl = leaves.size - 1
        ni = (nl - 1) // 7
        nn = nl + ni
        o = dict(
            num_leaves=nl, num_internal=ni, num_nodes=nn,
            prefixes=np.zeros(nn, dtype=leaves.dtype),
            child_offsets=np.zeros(nn + 1, dtype=np.int32),
            parents=np.zeros(max(1, (nn - 1) // 8), dtype=np.int32),
            level_range=np.zeros(max_level(kb) + 2, dtype=np.int32),
            internal_to_leaf=np.zeros(nn, dtype=np.int32),
            leaf_to_internal=np.zeros(nn, dtype=np.int32),
        )
        rc = self._f("build_octree")(C.c_int(kb), _p(leaves), C.c_int(nl), _p(o["prefixes"]), _p(o["child_offsets"]),
                                     _p(o["parents"]), _p(o["level_range"]), _p(o["internal_to_leaf"]),
                                     _p(o["leaf_to_internal"]))
        assert rc == 0, rc
        o["parents"] = o["parents"][:(nn - 1) // 8]
        return o

    def upsweep_counts(self, octree, leaf_counts):
        """leaf counts (cornerstone order) -> counts for all nodes of the linked octree"""
        nn = octree["num_nodes"]
        q = np.zeros(nn, dtype=np.uint32)
        leaf_nodes = octree["leaf_to_internal"][octree["num_internal"]:]
        q[leaf_nodes] = leaf_counts
        rc = self._f("upsweep_counts")(_p(octree["level_range"]), C.c_int(octree["level_range"].size),
                                       _p(octree["child_offsets"]), _p(q))
        assert rc == 0, rc
        return q

    def halo_radii(self, h, layout, first, last, num_leaves, ext=1.0):
        radii = np.zeros(num_leaves, dtype=np.float32)
        lay = np.ascontiguousarray(layout, dtype=np.uint32)
        rc = self._f("halo_radii")(C.c_int(h.dtype.itemsize * 8), _p(h), _p(lay), C.c_int(first), C.c_int(last),
                                   C.c_int(num_leaves), C.c_float(ext), _p(radii))
        assert rc == 0, rc
        return radii

    def find_halos(self, curve, octree, leaves, radii, box, first, last, real_bits=64):
        kb = leaves.dtype.itemsize * 8
        flags = np.zeros(leaves.size - 1, dtype=np.int32)
        radii = np.ascontiguousarray(radii, dtype=np.float32)
        rc = self._f("find_halos")(C.c_int(curve), C.c_int(kb), C.c_int(real_bits), _p(octree["prefixes"]),
                                   _p(octree["child_offsets"]), _p(octree["internal_to_leaf"]), _p(leaves),
                                   _p(radii), _p(box.lim), _p(box.bc), C.c_int(first), C.c_int(last), _p(flags))
        if rc == -3:
            return None
        assert rc == 0, rc
        return flags

    def node_centers(self, curve, prefixes, box, real_bits=64):
        kb = prefixes.dtype.itemsize * 8
        nn = prefixes.size
        centers = np.zeros((nn, 3), dtype=real_dtype(real_bits))
        sizes = np.zeros((nn, 3), dtype=real_dtype(real_bits))
        rc = self._f("node_centers")(C.c_int(curve), C.c_int(kb), C.c_int(real_bits), _p(prefixes), C.c_int(nn),
                                     _p(box.lim), _p(box.bc), _p(centers), _p(sizes))
        if rc == -3:
            return None
        assert rc == 0, rc
        return centers, sizes

    def find_neighbors(self, x, y, z, h, first, last, box, octree, layout, centers, sizes, ngmax, ext=1.0):
        nw = last - first
        nidx = np.zeros((nw, ngmax), dtype=np.uint32)
        nc = np.zeros(nw, dtype=np.uint32)
        lay = np.ascontiguousarray(layout, dtype=np.uint32)
        h = np.ascontiguousarray(h, dtype=x.dtype)
        rc = self._f("find_neighbors")(C.c_int(x.dtype.itemsize * 8), _p(x), _p(y), _p(z), _p(h), C.c_uint(first),
                                       C.c_uint(last), _p(box.lim), _p(box.bc), _p(octree["child_offsets"]),
                                       _p(octree["internal_to_leaf"]), _p(lay), _p(centers), _p(sizes),
                                       C.c_float(ext), C.c_uint(ngmax), _p(nidx), _p(nc))
        assert rc == 0, rc
        return nidx, nc

    # ---- target particle groups (oracle only: the reference's code for them is GPU-only) ----
    def fixed_groups(self, first, last, group_size):
        cap = (last - first + group_size - 1) // group_size + 1
        groups = np.zeros(cap, dtype=np.uint32)
        n = self._f("fixed_groups")(C.c_uint(first), C.c_uint(last), C.c_uint(group_size), _p(groups), C.c_int(cap))
        assert n >= 0, n
        return groups[:n + 1]

    def group_splits(self, first, last, x, y, z, leaves, layout, box, group_size, tol_factor):
        cap = last - first + 2
        groups = np.zeros(cap, dtype=np.uint32)
        lay = np.ascontiguousarray(layout, dtype=np.uint32)
        n = self._f("group_splits")(C.c_int(leaves.dtype.itemsize * 8), C.c_int(x.dtype.itemsize * 8), C.c_uint(first),
                                    C.c_uint(last), _p(x), _p(y), _p(z), _p(leaves), C.c_int(leaves.size - 1), _p(lay),
                                    _p(box.lim), _p(box.bc), C.c_uint(group_size), C.c_float(tol_factor), _p(groups),
                                    C.c_int(cap))
        assert n >= 0, n
        return groups[:n + 1]

    def find_splits(self, pos, dist_crit_sq):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        words = pos.shape[0] // 64
        out = np.zeros(words, dtype=np.uint64)
        self._f("find_splits")(_p(pos), C.c_int(words), C.c_double(dist_crit_sq), _p(out))
        return out

    def make_splits(self, masks, width=64):
        masks = np.ascontiguousarray(masks, dtype=np.uint64)
        out = np.zeros(masks.size * width + 1, dtype=np.uint32)
        n = self._f("make_splits")(_p(masks), C.c_int(masks.size), C.c_int(width), _p(out))
        return out[:n]

    def halo_boxes(self, curve, leaves, radii, box, first, last, real_bits=64):
        kb = leaves.dtype.itemsize * 8
        boxes = np.zeros((last - first, 8), dtype=np.int32)
        radii = np.ascontiguousarray(radii, dtype=np.float32)
        rc = self._f("halo_boxes")(C.c_int(curve), C.c_int(kb), C.c_int(real_bits), _p(leaves), _p(radii),
                                   _p(box.lim), _p(box.bc), C.c_int(first), C.c_int(last), _p(boxes))
        assert rc == 0, rc
        return boxes

    def find_overlaps(self, curve, leaves, boxes, first, last):
        kb = leaves.dtype.itemsize * 8
        flags = np.zeros(leaves.size - 1, dtype=np.int32)
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 8)
        rc = self._f("find_overlaps")(C.c_int(curve), C.c_int(kb), _p(leaves), _p(boxes), C.c_int(boxes.shape[0]),
                                      C.c_int(first), C.c_int(last), _p(flags))
        assert rc == 0, rc
        return flags

    # ---- focus tree (locally essential tree)
    def essential_ops(self, octree, counts, macs, focus_start, focus_end, bucket):
        pre = octree["prefixes"]
        nn = pre.size
        ops = np.zeros(nn, dtype=np.int32)
        rc = self._f("essential_ops")(C.c_int(pre.dtype.itemsize * 8), _p(pre), _p(octree["child_offsets"]),
                                      _p(octree["parents"]), _p(np.ascontiguousarray(counts, dtype=np.uint32)),
                                      _p(np.ascontiguousarray(macs, dtype=np.int8)), C.c_uint64(int(focus_start)),
                                      C.c_uint64(int(focus_end)), C.c_uint(bucket), _p(ops), C.c_int(nn))
        assert rc == 0, rc
        return ops

    def mac_refine_ops(self, octree, macs, num_leaves, focus_first, focus_last):
        pre = octree["prefixes"]
        ops = np.zeros(num_leaves, dtype=np.int32)
        # leaf index -> node index: the part of leafToInternal behind the internal nodes (R/tree/octree.hpp:367-370)
        l2i = np.ascontiguousarray(octree["leaf_to_internal"][octree["num_internal"]:])
        rc = self._f("mac_refine_ops")(C.c_int(pre.dtype.itemsize * 8), _p(pre),
                                       _p(np.ascontiguousarray(macs, dtype=np.int8)), _p(l2i),
                                       C.c_int(num_leaves), C.c_int(focus_first), C.c_int(focus_last), _p(ops))
        assert rc == 0, rc
        return ops

    def protect_ancestors(self, octree, ops):
        pre = octree["prefixes"]
        ops = np.ascontiguousarray(ops, dtype=np.int32).copy()
        rc = self._f("protect_ancestors")(C.c_int(pre.dtype.itemsize * 8), _p(pre), _p(octree["parents"]), _p(ops),
                                          C.c_int(pre.size))
        assert rc >= 0, rc
        return ops, bool(rc)

    def enforce_keys(self, forced_keys, octree, ops):
        pre = octree["prefixes"]
        ops = np.ascontiguousarray(ops, dtype=np.int32).copy()
        fk = np.ascontiguousarray(forced_keys, dtype=pre.dtype)
        rc = self._f("enforce_keys")(C.c_int(pre.dtype.itemsize * 8), _p(fk), C.c_int(fk.size), _p(pre),
                                     _p(octree["child_offsets"]), _p(octree["parents"]), _p(ops))
        assert rc >= 0, rc
        return ops, int(rc)

    def range_count(self, leaves, counts, leaves_focus, focus_idx, counts_focus=None):
        nf = leaves_focus.size - 1
        out = np.zeros(nf, dtype=np.uint32) if counts_focus is None else np.ascontiguousarray(counts_focus, dtype=np.uint32).copy()
        idx = np.ascontiguousarray(focus_idx, dtype=np.int32)
        rc = self._f("range_count")(C.c_int(leaves.dtype.itemsize * 8), _p(leaves), C.c_int(leaves.size - 1),
                                    _p(np.ascontiguousarray(counts, dtype=np.uint32)), _p(leaves_focus), C.c_int(nf),
                                    _p(idx), C.c_int(idx.size), _p(out))
        assert rc == 0, rc
        return out

    def mac_spheres(self, curve, mode, prefixes, box, inv_theta, real_bits=64, spheres=None):
        nn = prefixes.size
        sph = (np.zeros((nn, 4), dtype=real_dtype(real_bits)) if spheres is None
               else np.ascontiguousarray(spheres, dtype=real_dtype(real_bits)).copy())
        rc = self._f("mac_spheres")(C.c_int(curve), C.c_int(mode), C.c_int(prefixes.dtype.itemsize * 8),
                                    C.c_int(real_bits), _p(prefixes), C.c_int(nn), _p(sph), C.c_float(inv_theta),
                                    _p(box.lim), _p(box.bc))
        if rc == -3:
            return None
        assert rc == 0, rc
        return sph

    def mark_macs(self, curve, octree, centers, box, focus_nodes, limit_source, markings=None):
        pre = octree["prefixes"]
        real_bits = centers.dtype.itemsize * 8
        marks = np.zeros(pre.size, dtype=np.int8) if markings is None else np.ascontiguousarray(markings, dtype=np.int8).copy()
        fn = np.ascontiguousarray(focus_nodes, dtype=pre.dtype)
        rc = self._f("mark_macs")(C.c_int(curve), C.c_int(pre.dtype.itemsize * 8), C.c_int(real_bits), _p(pre),
                                  _p(octree["child_offsets"]), _p(np.ascontiguousarray(centers)), _p(box.lim),
                                  _p(box.bc), _p(fn), C.c_int(fn.size - 1), C.c_int(int(limit_source)), _p(marks))
        if rc == -3:
            return None
        assert rc == 0, rc
        return marks

    def span_sfc_range(self, key_bits, a, b):
        f = self._f("span_sfc_range")
        n = f(C.c_int(key_bits), C.c_uint64(int(a)), C.c_uint64(int(b)), None)
        assert n >= 0, n
        out = np.zeros(n, dtype=key_dtype(key_bits))
        if n:
            n2 = f(C.c_int(key_bits), C.c_uint64(int(a)), C.c_uint64(int(b)), _p(out))
            assert n2 == n
        return out

    def leaf_source_centers(self, x, y, z, m, leaf_to_internal, layout, num_nodes, center_bits=64):
        ctr = np.zeros((num_nodes, 4), dtype=real_dtype(center_bits))
        l2i = np.ascontiguousarray(leaf_to_internal, dtype=np.int32)
        lay = np.ascontiguousarray(layout, dtype=np.uint32)
        rc = self._f("leaf_source_centers")(C.c_int(x.dtype.itemsize * 8), C.c_int(m.dtype.itemsize * 8),
                                            C.c_int(center_bits), _p(x), _p(y), _p(z), _p(m), _p(l2i),
                                            C.c_int(lay.size - 1), _p(lay), _p(ctr))
        assert rc == 0, rc
        return ctr

    def upsweep_centers(self, octree, centers, num_levels):
        ctr = np.ascontiguousarray(centers).copy()
        rc = self._f("upsweep_centers")(C.c_int(ctr.dtype.itemsize * 8), C.c_int(num_levels), _p(octree["level_range"]),
                                        _p(octree["child_offsets"]), _p(ctr))
        assert rc == 0, rc
        return ctr

    def segment_max(self, values, segments, out_bits):
        seg = np.ascontiguousarray(segments, dtype=np.uint32)
        out = np.zeros(seg.size - 1, dtype=real_dtype(out_bits))
        rc = self._f("segment_max")(C.c_int(values.dtype.itemsize * 8), C.c_int(out_bits), _p(values), _p(seg),
                                    C.c_size_t(seg.size - 1), _p(out))
        assert rc == 0, rc
        return out

    def binary_tree(self, leaves):
        """(child[n][2], prefix[n]) of the binary radix tree over a cornerstone leaf array (btree)"""
        n = leaves.size - 1
        child = np.zeros((n, 2), dtype=np.int32)
        prefix = np.zeros(n, dtype=leaves.dtype)
        rc = self._f("binary_tree")(C.c_int(leaves.dtype.itemsize * 8), _p(leaves), C.c_int(n), _p(child), _p(prefix))
        assert rc == 0, rc
        return child, prefix

    def num_threads(self):
        return int(self._f("num_threads")())


class Oracle(_CpuImpl):
    prefix = "cstone_oracle_"
    # (CSTONE_ORACLE_LIB: the sanitizer build of `make -C oracle asan`)
    libpath = os.environ.get("CSTONE_ORACLE_LIB", os.path.join(HERE, "libcstone_oracle.so"))

    def random_uniform(self, n, box, seed=42, real_bits=64):
        """the reference's RandomCoordinates cloud (std::mt19937(seed); all x, then all y, then all z)"""
        x, y, z = [np.empty(n, dtype=real_dtype(real_bits)) for _ in range(3)]
        rc = self._f("random_uniform")(C.c_int(real_bits), C.c_uint(seed), C.c_size_t(n), _p(box.lim), _p(x), _p(y), _p(z))
        assert rc == 0, rc
        return x, y, z

    def plummer(self, n, real_bits=64):
        """the reference's Plummer sphere (test/coord_samples/plummer.hpp, srand48(42)), serial"""
        x, y, z = [np.empty(n, dtype=real_dtype(real_bits)) for _ in range(3)]
        rc = self._f("plummer")(C.c_int(real_bits), C.c_size_t(n), _p(x), _p(y), _p(z))
        assert rc == 0, rc
        return x, y, z


class Reference(_CpuImpl):
    prefix = "cstone_ref_"
    libpath = os.path.join(HERE, "_ref", "libcstone_ref.so")

    plummer = Oracle.plummer  # (cstone_ref_plummer: the reference's own plummer<T>(n))


def reference_available():
    return os.path.exists(Reference.libpath)
