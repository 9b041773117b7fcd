"""Python plumbing for libcstone_hip.so (ctypes over the C ABI of include/cstone_hip.h).

This is NOT the product: the product is the shared library and the C++20 headers in
cornerstone-octree_amd/include/.  The module exists so that tests/, bench.py and smoke() can drive
the C ABI with torch tensors as device buffers (torch = device memory + streams + torch.distributed).
There is NO fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIBPATH = os.path.join(PKG, "lib", "libcstone_hip.so")

MORTON, HILBERT = 0, 1

STAGES = {
    "encode": 0, "sort_hist": 1, "sort_pass": 2, "gather": 3, "node_counts": 4, "rebalance": 5, "link_octree": 6,
    "halos": 7, "neighbors": 8, "minmax": 9, "sort_pass_iota": 10, "resort_bins": 11, "resort_leaves": 12, "gather_h": 13, "place": 14,
}

EXPORTS = [
    "cstone_hip_ctx_create", "cstone_hip_ctx_destroy", "cstone_hip_ctx_sync", "cstone_hip_last_error",
    "cstone_hip_device_info", "cstone_hip_malloc", "cstone_hip_free", "cstone_hip_memcpy_h2d",
    "cstone_hip_memcpy_d2h", "cstone_hip_memcpy_d2d", "cstone_hip_memset", "cstone_hip_profile_enable", "cstone_hip_profile_markers", "cstone_hip_domain_mr_sync_grav", "cstone_hip_domain_mr_update_expansion_centers", "cstone_hip_add_macs", "cstone_hip_halo_request_rows", "cstone_hip_peer_range_counts", "cstone_hip_adjacent_difference_u32", "cstone_hip_offsets_from_counts_u32", "cstone_hip_domain_sync_grav", "cstone_hip_domain_update_expansion_centers", "cstone_hip_gather_tables_u32",
    "cstone_hip_profile_reset", "cstone_hip_profile_get", "cstone_hip_profile_get_spread", "cstone_hip_compute_sfc_keys",
    "cstone_hip_sort_pairs_temp_bytes", "cstone_hip_sort_pairs", "cstone_hip_sort_keys_ordering", "cstone_hip_sfc_keys_and_ordering", "cstone_hip_sequence_u32", "cstone_hip_gather",
    "cstone_hip_scatter", "cstone_hip_gather_scatter", "cstone_hip_merge_positions", "cstone_hip_minmax", "cstone_hip_minmax_arrays", "cstone_hip_exclusive_scan_u32", "cstone_hip_inclusive_scan_u32", "cstone_hip_lower_bound",
    "cstone_hip_compute_node_counts", "cstone_hip_compute_node_counts_guided", "cstone_hip_compute_node_ops", "cstone_hip_rebalance_tree",
    "cstone_hip_update_octree", "cstone_hip_compute_octree", "cstone_hip_build_octree", "cstone_hip_upsweep_sum",
    "cstone_hip_node_centers", "cstone_hip_halo_radii", "cstone_hip_find_halos", "cstone_hip_find_neighbors",
    "cstone_hip_halo_boxes", "cstone_hip_halo_boxes_foreign", "cstone_hip_find_overlaps", "cstone_hip_compute_fixed_groups",
    "cstone_hip_compute_group_splits", "cstone_hip_find_neighbors_groups",
    "cstone_hip_domain_create", "cstone_hip_domain_destroy", "cstone_hip_domain_sync", "cstone_hip_domain_view_get",
    "cstone_hip_domain_set_halo_factor", "cstone_hip_domain_mr_create", "cstone_hip_domain_mr_destroy",
    "cstone_hip_domain_mr_sync", "cstone_hip_domain_mr_sync_props", "cstone_hip_domain_mr_sync_keys", "cstone_hip_domain_mr_view_get", "cstone_hip_domain_mr_set_halo_factor", "cstone_hip_domain_mr_exchange_halos", "cstone_hip_domain_mr_reapply_sync",
    "cstone_hip_domain_reapply_sync", "cstone_hip_domain_stats_get", "cstone_hip_domain_mr_octree_get",
    "cstone_hip_fill", "cstone_hip_scale", "cstone_hip_increment", "cstone_hip_count_equal", "cstone_hip_reduce_sum",
    "cstone_hip_max_norm_square", "cstone_hip_segment_max", "cstone_hip_gather_ranges", "cstone_hip_lower_bound_value",
    "cstone_hip_sort_keys", "cstone_hip_rebalance_decision_essential", "cstone_hip_mac_refine_decision",
    "cstone_hip_protect_ancestors", "cstone_hip_enforce_keys", "cstone_hip_range_count", "cstone_hip_mark_macs",
    "cstone_hip_count_sfc_gaps", "cstone_hip_fill_sfc_gaps", "cstone_hip_geo_mac_spheres", "cstone_hip_set_mac",
    "cstone_hip_move_centers", "cstone_hip_leaf_source_centers", "cstone_hip_upsweep_centers",
    "cstone_hip_comm_rccl_unique_id", "cstone_hip_comm_rccl_create", "cstone_hip_comm_rccl_ops",
    "cstone_hip_comm_rccl_destroy", "cstone_hip_create_binary_tree",
    # the device work behind the locally essential tree's host state machine (csrc/let.hpp)
    "cstone_hip_raise", "cstone_hip_find_peers_mac", "cstone_hip_keys_missing", "cstone_hip_partition_keys",
    "cstone_hip_zero_ops_at_keys", "cstone_hip_locate_nodes", "cstone_hip_node_layout", "cstone_hip_halo_requests",
    "cstone_hip_ranges_from_keys", "cstone_hip_domain_mr_set_halo_mode", "cstone_hip_domain_mr_set_theta",
    "cstone_hip_upload", "cstone_hip_focus_update_ops", "cstone_hip_find_neighbors_stats",
    "cstone_hip_gather_multi", "cstone_hip_domain_sync_scratch", "cstone_hip_gather_ranges_rows",
    "cstone_hip_scatter_rows", "cstone_hip_build_octree_bounded", "cstone_hip_upsweep_sum_bounded",
    "cstone_hip_lower_bound_u32", "cstone_hip_sequence_u64", "cstone_hip_scan_u32_to_u64",
    "cstone_hip_domain_set_sort_mode", "cstone_hip_domain_set_speculative_box", "cstone_hip_test_hooks",
    "cstone_hip_domain_mr_set_sort_mode", "cstone_hip_find_neighbors_interleaved",
]


class CBox(C.Structure):
    _fields_ = [("lim", C.c_double * 6), ("bc", C.c_int32 * 3), ("pad_", C.c_int32)]


def make_cbox(lim, bc=(0, 0, 0)):
    b = CBox()
    for i in range(6):
        b.lim[i] = float(lim[i])
    for i in range(3):
        b.bc[i] = int(bc[i])
    return b


class CstoneError(RuntimeError):
    pass


def max_level(key_bits):
    return {32: 10, 64: 21}[key_bits]


def load_library():
    """dlopen the product library; raises if it has not been built (no fallback of any kind)"""
    path = os.environ.get("CSTONE_HIP_LIB", LIBPATH)  # tuning builds; the default is the in-tree product library
    if not os.path.exists(path):
        raise CstoneError(f"{LIBPATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"or `make -C cornerstone-octree_amd`")
    # torch first: it brings its own libamdhip64, and a process must hold ONE HIP runtime -- with the library's
    # /opt/rocm copy loaded before torch's, the second runtime finds no device ("no ROCm-capable device is detected")
    import torch  # noqa: F401

    lib = C.CDLL(path)
    lib.cstone_hip_last_error.restype = C.c_char_p
    lib.cstone_hip_sort_pairs_temp_bytes.restype = C.c_size_t
    return lib


def _torch():
    import torch

    return torch


def _ptr(t):
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def key_torch_dtype(key_bits):
    torch = _torch()
    # torch has no unsigned 32/64 arithmetic types we need; keys live in int32/int64 storage (bit patterns)
    return {32: torch.int32, 64: torch.int64}[key_bits]


def keys_to_numpy(t, key_bits):
    a = t.cpu().numpy()
    return a.view(np.uint32 if key_bits == 32 else np.uint64)


def keys_from_numpy(a, device):
    torch = _torch()
    a = np.ascontiguousarray(a)
    signed = a.view(np.int32 if a.dtype.itemsize == 4 else np.int64)
    return torch.from_numpy(signed.copy()).to(device)


class Context:
    """One cstone_hip context bound to a torch device and (by default) torch's current stream."""

    def __init__(self, device=0, use_torch_stream=True):
        torch = _torch()
        self.lib = load_library()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if not torch.cuda.is_available():
            raise CstoneError("no GPU visible: libcstone_hip needs an MI355X (there is no CPU path)")
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream if use_torch_stream else 0
        # the stream the library's work is ordered on, as torch sees it (None: a private stream of the library)
        self.stream_handle = stream if use_torch_stream else None
        self.h = C.c_void_p()
        rc = self.lib.cstone_hip_ctx_create(C.byref(self.h), C.c_int(self.device.index), C.c_void_p(stream),
                                            C.c_int(0 if use_torch_stream else 1))
        if rc != 0:
            self.lib.cstone_hip_last_error.restype = C.c_char_p
            why = self.lib.cstone_hip_last_error(C.c_void_p(None))
            raise CstoneError(f"cstone_hip_ctx_create failed: {rc} ({why.decode() if why else '?'})")

    def close(self):
        if getattr(self, "h", None):
            self.lib.cstone_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        # not during interpreter shutdown: the HIP runtime may be gone by then (objects kept alive by a traceback are
        # collected that late), and the process is about to release everything anyway
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.cstone_hip_last_error(self.h)
            raise CstoneError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def sync(self):
        self._chk(self.lib.cstone_hip_ctx_sync(self.h), "ctx_sync")

    # ---- profiling
    def profile_enable(self, on=True):
        """True / 1: every stage; 2: only the stages of the kernels that move the particle arrays; False: off"""
        self._chk(self.lib.cstone_hip_profile_enable(self.h, C.c_int(int(on))), "profile_enable")

    def profile_markers(self, on=True):
        """roctx ranges "cstone:<stage>" around every stage (rocprofv3 --marker-trace)"""
        self._chk(self.lib.cstone_hip_profile_markers(self.h, C.c_int(int(on))), "profile_markers")

    def profile_reset(self):
        self._chk(self.lib.cstone_hip_profile_reset(self.h), "profile_reset")

    def profile_get(self, stage):
        ms, cnt = C.c_double(0), C.c_int(0)
        self._chk(self.lib.cstone_hip_profile_get(self.h, C.c_int(STAGES[stage]), C.byref(ms), C.byref(cnt)),
                  "profile_get")
        return ms.value, cnt.value

    def profile_spread(self, stage):
        """(min, median, max) ms of the individual brackets of a stage since the last reset"""
        lo, med, hi = C.c_double(0), C.c_double(0), C.c_double(0)
        self._chk(self.lib.cstone_hip_profile_get_spread(self.h, C.c_int(STAGES[stage]), C.byref(lo), C.byref(med),
                                                         C.byref(hi)), "profile_get_spread")
        return lo.value, med.value, hi.value

    # ---- keys
    def compute_sfc_keys(self, curve, key_bits, x, y, z, box, keys=None):
        torch = _torch()
        n = x.numel()
        if keys is None:
            keys = torch.zeros(n, dtype=key_torch_dtype(key_bits), device=x.device)
        rb = x.element_size() * 8
        self._chk(self.lib.cstone_hip_compute_sfc_keys(self.h, C.c_int(curve), C.c_int(key_bits), C.c_int(rb),
                                                       _ptr(x), _ptr(y), _ptr(z), _ptr(keys), C.c_size_t(n),
                                                       C.byref(box)), "compute_sfc_keys")
        return keys

    # ---- sort
    def sort_temp_bytes(self, key_bits, n):
        return int(self.lib.cstone_hip_sort_pairs_temp_bytes(C.c_int(key_bits), C.c_size_t(n)))

    def sort_pairs(self, keys, vals, keys_alt=None, vals_alt=None, temp=None):
        """in place; vals int32 storage holding uint32"""
        kb = keys.element_size() * 8
        n = keys.numel()
        tb = temp.numel() * temp.element_size() if temp is not None else 0
        self._chk(self.lib.cstone_hip_sort_pairs(self.h, C.c_int(kb), _ptr(keys), _ptr(vals), C.c_size_t(n),
                                                 _ptr(keys_alt), _ptr(vals_alt), _ptr(temp), C.c_size_t(tb)),
                  "sort_pairs")

    def sort_keys_ordering(self, keys):
        """sorts keys in place, returns the sorting permutation (int32 storage)"""
        torch = _torch()
        kb, n = keys.element_size() * 8, keys.numel()
        order = torch.empty(n, dtype=torch.int32, device=keys.device)
        ka, va = torch.empty_like(keys), torch.empty_like(order)
        tmp = torch.empty(max(1, self.sort_temp_bytes(kb, n)), dtype=torch.uint8, device=keys.device)
        self._chk(self.lib.cstone_hip_sort_keys_ordering(self.h, C.c_int(kb), _ptr(keys), _ptr(order), C.c_size_t(n),
                                                         _ptr(ka), _ptr(va), _ptr(tmp),
                                                         C.c_size_t(tmp.numel())), "sort_keys_ordering")
        return order

    def sfc_keys_and_ordering(self, curve, key_bits, x, y, z, box, keys=None):
        """(sorted keys, ordering): compute_sfc_keys + sort_keys_ordering in one call"""
        torch = _torch()
        n = x.numel()
        if keys is None:
            keys = torch.zeros(n, dtype=key_torch_dtype(key_bits), device=x.device)
        order = torch.empty(n, dtype=torch.int32, device=x.device)
        ka, va = torch.empty_like(keys), torch.empty_like(order)
        tmp = torch.empty(max(1, self.sort_temp_bytes(key_bits, n)), dtype=torch.uint8, device=x.device)
        self._chk(self.lib.cstone_hip_sfc_keys_and_ordering(self.h, C.c_int(curve), C.c_int(key_bits),
                                                            C.c_int(x.element_size() * 8), _ptr(x), _ptr(y), _ptr(z),
                                                            _ptr(keys), _ptr(order), C.c_size_t(n), C.byref(box),
                                                            _ptr(ka), _ptr(va), _ptr(tmp), C.c_size_t(tmp.numel())),
                  "sfc_keys_and_ordering")
        return keys, order

    def sequence(self, out, init=0):
        self._chk(self.lib.cstone_hip_sequence_u32(self.h, _ptr(out), C.c_size_t(out.numel()), C.c_uint32(init)),
                  "sequence_u32")

    def gather(self, map_, src, dst, elem_bytes=None, n=None):
        eb = elem_bytes or src.element_size()
        n = map_.numel() if n is None else n
        self._chk(self.lib.cstone_hip_gather(self.h, C.c_int(eb), _ptr(map_), C.c_size_t(n), _ptr(src), _ptr(dst)),
                  "gather")

    def scatter(self, map_, src, dst, elem_bytes=None):
        eb = elem_bytes or src.element_size()
        self._chk(self.lib.cstone_hip_scatter(self.h, C.c_int(eb), _ptr(map_), C.c_size_t(map_.numel()), _ptr(src),
                                              _ptr(dst)), "scatter")

    def gather_scatter(self, map_in, map_out, src, dst):
        """dst[map_out[i]] = src[map_in[i]]"""
        self._chk(self.lib.cstone_hip_gather_scatter(self.h, C.c_int(src.element_size()), _ptr(map_in), _ptr(map_out),
                                                     C.c_size_t(map_in.numel()), _ptr(src), _ptr(dst)),
                  "gather_scatter")

    def merge_positions(self, keys_a, keys_b, offset=0):
        """positions (int32 storage) of two sorted key runs in their stable merge, ties: run a first"""
        torch = _torch()
        pa = torch.empty(keys_a.numel(), dtype=torch.int32, device=keys_a.device)
        pb = torch.empty(keys_b.numel(), dtype=torch.int32, device=keys_a.device)
        self._chk(self.lib.cstone_hip_merge_positions(self.h, C.c_int(keys_a.element_size() * 8), _ptr(keys_a),
                                                      C.c_size_t(keys_a.numel()), _ptr(keys_b),
                                                      C.c_size_t(keys_b.numel()), C.c_uint32(offset), _ptr(pa),
                                                      _ptr(pb)), "merge_positions")
        return pa, pb

    def minmax(self, x):
        out = (C.c_double * 2)()
        self._chk(self.lib.cstone_hip_minmax(self.h, C.c_int(x.element_size() * 8), _ptr(x), C.c_size_t(x.numel()),
                                             out), "minmax")
        return out[0], out[1]

    def minmax_arrays(self, arrays):
        """[(min, max), ...] of 1..3 equally long arrays: one launch, one read-back"""
        k = len(arrays)
        out = (C.c_double * (2 * k))()
        ptrs = (C.c_void_p * k)(*[a.data_ptr() for a in arrays])
        self._chk(self.lib.cstone_hip_minmax_arrays(self.h, C.c_int(arrays[0].element_size() * 8), ptrs, C.c_int(k),
                                                    C.c_size_t(arrays[0].numel()), out), "minmax_arrays")
        return [(out[2 * i], out[2 * i + 1]) for i in range(k)]

    def exclusive_scan(self, inp, out, init=0):
        self._chk(self.lib.cstone_hip_exclusive_scan_u32(self.h, _ptr(inp), _ptr(out), C.c_size_t(inp.numel()),
                                                         C.c_uint32(init)), "exclusive_scan")

    def inclusive_scan(self, inp, out):
        self._chk(self.lib.cstone_hip_inclusive_scan_u32(self.h, _ptr(inp), _ptr(out), C.c_size_t(inp.numel())),
                  "inclusive_scan")

    def offsets_from_counts(self, counts, out):
        """out[i] = sum(counts[0..i)) for i <= n (out holds n + 1 values)"""
        assert out.numel() == counts.numel() + 1
        self._chk(self.lib.cstone_hip_offsets_from_counts_u32(self.h, _ptr(counts), _ptr(out), C.c_size_t(counts.numel())),
                  "offsets_from_counts")

    def gather_tables(self, index_map, a, n_a, b, n_b, c, out):
        """out = [a[map[:n_a]] (zeros if a is None), b[map[n_a:n_a + n_b]], c] (uint32 device tensors)"""
        n_c = 0 if c is None else c.numel()
        self._chk(self.lib.cstone_hip_gather_tables_u32(self.h, _ptr(index_map), C.c_void_p(a.data_ptr() if a is not None else 0),
                                                        C.c_size_t(n_a), _ptr(b), C.c_size_t(n_b),
                                                        C.c_void_p(c.data_ptr() if c is not None else 0), C.c_size_t(n_c),
                                                        _ptr(out)), "gather_tables")

    def lower_bound(self, keys, values):
        """positions (int64 device tensor) of the first key >= values[q], unsigned comparison of the bit patterns"""
        torch = _torch()
        out = torch.empty(values.numel(), dtype=torch.int64, device=keys.device)
        self._chk(self.lib.cstone_hip_lower_bound(self.h, C.c_int(keys.element_size() * 8), _ptr(keys),
                                                  C.c_size_t(keys.numel()), _ptr(values), C.c_int(values.numel()),
                                                  _ptr(out)), "lower_bound")
        return out

    # ---- tree
    def compute_node_counts(self, tree, keys, counts=None, max_count=0xFFFFFFFF, num_nodes=None):
        torch = _torch()
        kb = keys.element_size() * 8
        nn = tree.numel() - 1 if num_nodes is None else num_nodes
        if counts is None:
            counts = torch.zeros(nn, dtype=torch.int32, device=tree.device)
        self._chk(self.lib.cstone_hip_compute_node_counts(self.h, C.c_int(kb), _ptr(tree), _ptr(counts), C.c_int(nn),
                                                          _ptr(keys), C.c_size_t(keys.numel()),
                                                          C.c_uint32(max_count)), "compute_node_counts")
        return counts

    def compute_node_counts_guided(self, tree, keys, guess, max_count=0xFFFFFFFF):
        torch = _torch()
        nn = tree.numel() - 1
        counts = torch.zeros(nn, dtype=torch.int32, device=tree.device)
        self._chk(self.lib.cstone_hip_compute_node_counts_guided(self.h, C.c_int(keys.element_size() * 8), _ptr(tree),
                                                                 _ptr(counts), C.c_int(nn), _ptr(keys),
                                                                 C.c_size_t(keys.numel()), C.c_uint32(max_count),
                                                                 _ptr(guess)), "compute_node_counts_guided")
        return counts

    def compute_node_ops(self, tree, counts, bucket, num_nodes=None):
        torch = _torch()
        kb = tree.element_size() * 8
        nn = tree.numel() - 1 if num_nodes is None else num_nodes
        ops = torch.zeros(nn + 1, dtype=torch.int32, device=tree.device)
        new_n, conv = C.c_int(0), C.c_int(0)
        self._chk(self.lib.cstone_hip_compute_node_ops(self.h, C.c_int(kb), _ptr(tree), C.c_int(nn), _ptr(counts),
                                                       C.c_uint32(bucket), _ptr(ops), C.byref(new_n), C.byref(conv)),
                  "compute_node_ops")
        return ops, new_n.value, bool(conv.value)

    def rebalance_tree(self, tree, ops, new_num_nodes, num_nodes=None):
        torch = _torch()
        kb = tree.element_size() * 8
        nn = tree.numel() - 1 if num_nodes is None else num_nodes
        new_tree = torch.zeros(new_num_nodes + 1, dtype=tree.dtype, device=tree.device)
        self._chk(self.lib.cstone_hip_rebalance_tree(self.h, C.c_int(kb), _ptr(tree), C.c_int(nn),
                                                     C.c_int(new_num_nodes), _ptr(ops), _ptr(new_tree)),
                  "rebalance_tree")
        return new_tree

    def update_octree(self, keys, bucket, tree_buf, counts_buf, num_leaves, max_count=0xFFFFFFFF):
        """tree_buf / counts_buf are capacity buffers; returns (num_leaves, converged)"""
        kb = keys.element_size() * 8
        cap = counts_buf.numel()
        assert tree_buf.numel() >= cap + 1
        nl, conv = C.c_int(num_leaves), C.c_int(0)
        rc = self.lib.cstone_hip_update_octree(self.h, C.c_int(kb), _ptr(keys), C.c_size_t(keys.numel()),
                                               C.c_uint32(bucket), _ptr(tree_buf), _ptr(counts_buf), C.byref(nl),
                                               C.c_int(cap), C.c_uint32(max_count), C.byref(conv))
        if rc == -2:
            return -nl.value, False
        self._chk(rc, "update_octree")
        return nl.value, bool(conv.value)

    def compute_octree(self, keys, bucket, cap_leaves=None, max_count=0xFFFFFFFF):
        torch = _torch()
        kb = keys.element_size() * 8
        n = keys.numel()
        cap = cap_leaves or max(4096, 4 * n // max(1, bucket) + 4096)
        while True:
            tree = torch.zeros(cap + 1, dtype=keys.dtype, device=keys.device)
            counts = torch.zeros(cap, dtype=torch.int32, device=keys.device)
            nl, iters = C.c_int(0), C.c_int(0)
            rc = self.lib.cstone_hip_compute_octree(self.h, C.c_int(kb), _ptr(keys), C.c_size_t(n), C.c_uint32(bucket),
                                                    _ptr(tree), _ptr(counts), C.byref(nl), C.c_int(cap),
                                                    C.c_uint32(max_count), C.byref(iters))
            if rc == -2:
                cap = nl.value + 1
                continue
            self._chk(rc, "compute_octree")
            return tree[:nl.value + 1], counts[:nl.value], iters.value

    def build_octree(self, leaves, num_leaves=None):
        torch = _torch()
        kb = leaves.element_size() * 8
        nl = leaves.numel() - 1 if num_leaves is None else num_leaves
        ni = (nl - 1) // 7
        nn = nl + ni
        dev = leaves.device
        o = dict(
            num_leaves=nl, num_internal=ni, num_nodes=nn,
            prefixes=torch.zeros(nn, dtype=leaves.dtype, device=dev),
            child_offsets=torch.zeros(nn + 1, dtype=torch.int32, device=dev),
            parents=torch.zeros(max(1, (nn - 1) // 8), dtype=torch.int32, device=dev),
            level_range=torch.zeros(max_level(kb) + 2, dtype=torch.int32, device=dev),
            internal_to_leaf=torch.zeros(nn, dtype=torch.int32, device=dev),
            leaf_to_internal=torch.zeros(nn, dtype=torch.int32, device=dev),
        )
        self._chk(self.lib.cstone_hip_build_octree(self.h, C.c_int(kb), _ptr(leaves), C.c_int(nl), _ptr(o["prefixes"]),
                                                   _ptr(o["child_offsets"]), _ptr(o["parents"]),
                                                   _ptr(o["level_range"]), _ptr(o["internal_to_leaf"]),
                                                   _ptr(o["leaf_to_internal"])), "build_octree")
        return o

    def upsweep_sum(self, octree, counts):
        self._chk(self.lib.cstone_hip_upsweep_sum(self.h, C.c_int(octree["level_range"].numel()),
                                                  _ptr(octree["level_range"]), _ptr(octree["child_offsets"]),
                                                  _ptr(counts)), "upsweep_sum")

    def node_centers(self, curve, prefixes, box, real_bits=64):
        torch = _torch()
        kb = prefixes.element_size() * 8
        nn = prefixes.numel()
        dt = torch.float32 if real_bits == 32 else torch.float64
        centers = torch.zeros((nn, 3), dtype=dt, device=prefixes.device)
        sizes = torch.zeros((nn, 3), dtype=dt, device=prefixes.device)
        self._chk(self.lib.cstone_hip_node_centers(self.h, C.c_int(curve), C.c_int(kb), C.c_int(real_bits),
                                                   _ptr(prefixes), C.c_int(nn), C.byref(box), _ptr(centers),
                                                   _ptr(sizes)), "node_centers")
        return centers, sizes

    # ---- halos
    def halo_radii(self, h, layout, first, last, num_leaves, ext=1.0):
        torch = _torch()
        radii = torch.empty(num_leaves, dtype=torch.float32, device=h.device)
        self._chk(self.lib.cstone_hip_halo_radii(self.h, C.c_int(h.element_size() * 8), _ptr(h), _ptr(layout),
                                                 C.c_int(first), C.c_int(last), C.c_int(num_leaves), C.c_float(ext),
                                                 _ptr(radii)), "halo_radii")
        return radii

    def find_halos(self, curve, octree, leaves, radii, box, first, last, real_bits=64, flags=None):
        torch = _torch()
        kb = leaves.element_size() * 8
        if flags is None:
            flags = torch.zeros(octree["num_leaves"], dtype=torch.int32, device=leaves.device)
        self._chk(self.lib.cstone_hip_find_halos(self.h, C.c_int(curve), C.c_int(kb), C.c_int(real_bits),
                                                 _ptr(octree["prefixes"]), _ptr(octree["child_offsets"]),
                                                 _ptr(octree["internal_to_leaf"]), _ptr(leaves), _ptr(radii),
                                                 C.byref(box), C.c_int(first), C.c_int(last), _ptr(flags)),
                  "find_halos")
        return flags

    def halo_boxes(self, curve, leaves, radii, box, first, last, real_bits=64):
        torch = _torch()
        kb = leaves.element_size() * 8
        boxes = torch.zeros((last - first, 8), dtype=torch.int32, device=leaves.device)
        self._chk(self.lib.cstone_hip_halo_boxes(self.h, C.c_int(curve), C.c_int(kb), C.c_int(real_bits), _ptr(leaves),
                                                 _ptr(radii), C.byref(box), C.c_int(first), C.c_int(last),
                                                 _ptr(boxes)), "halo_boxes")
        return boxes

    def halo_boxes_foreign(self, curve, octree, leaves, radii, box, first, last, real_bits=64):
        """halo_boxes whose record[6] is set only for boxes that really overlap a leaf outside [first, last)"""
        torch = _torch()
        kb = leaves.element_size() * 8
        boxes = torch.zeros((last - first, 8), dtype=torch.int32, device=leaves.device)
        self._chk(self.lib.cstone_hip_halo_boxes_foreign(
            self.h, C.c_int(curve), C.c_int(kb), C.c_int(real_bits), _ptr(octree["prefixes"]),
            _ptr(octree["child_offsets"]), _ptr(octree["internal_to_leaf"]), _ptr(leaves), _ptr(radii), C.byref(box),
            C.c_int(first), C.c_int(last), _ptr(boxes)), "halo_boxes_foreign")
        return boxes

    def find_overlaps(self, curve, octree, leaves, boxes, first, last, flags=None):
        torch = _torch()
        kb = leaves.element_size() * 8
        if flags is None:
            flags = torch.zeros(octree["num_leaves"], dtype=torch.int32, device=leaves.device)
        nb = boxes.shape[0] if boxes.dim() == 2 else boxes.numel() // 8
        self._chk(self.lib.cstone_hip_find_overlaps(self.h, C.c_int(curve), C.c_int(kb), _ptr(octree["prefixes"]),
                                                    _ptr(octree["child_offsets"]), _ptr(octree["internal_to_leaf"]),
                                                    _ptr(leaves), _ptr(boxes), C.c_int(nb), C.c_int(first),
                                                    C.c_int(last), _ptr(flags)), "find_overlaps")
        return flags

    def find_neighbors(self, x, y, z, h, first, last, box, octree, layout, centers, sizes, ngmax, ext=1.0):
        torch = _torch()
        nw = last - first
        nidx = torch.zeros((nw, ngmax), dtype=torch.int32, device=x.device)
        nc = torch.zeros(nw, dtype=torch.int32, device=x.device)
        self._chk(self.lib.cstone_hip_find_neighbors(self.h, C.c_int(x.element_size() * 8), _ptr(x), _ptr(y), _ptr(z),
                                                     _ptr(h), C.c_uint32(first), C.c_uint32(last), C.byref(box),
                                                     _ptr(octree["child_offsets"]), _ptr(octree["internal_to_leaf"]),
                                                     _ptr(layout), _ptr(centers), _ptr(sizes), C.c_float(ext),
                                                     C.c_uint32(ngmax), _ptr(nidx), _ptr(nc)), "find_neighbors")
        return nidx, nc

    def compute_fixed_groups(self, first, last, group_size):
        torch = _torch()
        cap = (last - first + group_size - 1) // max(group_size, 1) + 1
        groups = torch.zeros(cap, dtype=torch.int32, device=self.device)
        ng = C.c_uint32(0)
        self._chk(self.lib.cstone_hip_compute_fixed_groups(self.h, C.c_uint32(first), C.c_uint32(last),
                                                           C.c_uint32(group_size), _ptr(groups), C.byref(ng)),
                  "compute_fixed_groups")
        return groups[:ng.value + 1]

    def compute_group_splits(self, first, last, x, y, z, leaves, layout, box, group_size, tol_factor, capacity=None):
        """groups[num_groups + 1]: groupStart = groups[:-1], groupEnd = groups[1:]"""
        torch = _torch()
        cap = last - first + 1 if capacity is None else capacity
        groups = torch.zeros(max(cap, 1), dtype=torch.int32, device=self.device)
        ng = C.c_uint32(0)
        self._chk(self.lib.cstone_hip_compute_group_splits(
            self.h, C.c_int(leaves.element_size() * 8), C.c_int(x.element_size() * 8), C.c_uint32(first),
            C.c_uint32(last), _ptr(x), _ptr(y), _ptr(z), _ptr(leaves), C.c_int(leaves.numel() - 1), _ptr(layout),
            C.byref(box), C.c_uint32(group_size), C.c_float(tol_factor), _ptr(groups), C.c_size_t(cap), C.byref(ng)),
            "compute_group_splits")
        return groups[:ng.value + 1]

    def find_neighbors_groups(self, x, y, z, h, first, last, groups, box, octree, layout, centers, sizes, ngmax,
                              ext=1.0):
        torch = _torch()
        nw = last - first
        nidx = torch.zeros((nw, ngmax), dtype=torch.int32, device=x.device)
        nc = torch.zeros(nw, dtype=torch.int32, device=x.device)
        ng = groups.numel() - 1
        self._chk(self.lib.cstone_hip_find_neighbors_groups(
            self.h, C.c_int(x.element_size() * 8), _ptr(x), _ptr(y), _ptr(z), _ptr(h), C.c_uint32(first),
            C.c_uint32(last), _ptr(groups), C.c_void_p(groups.data_ptr() + 4), C.c_uint32(ng), C.byref(box),
            _ptr(octree["child_offsets"]), _ptr(octree["internal_to_leaf"]), _ptr(layout), _ptr(centers), _ptr(sizes),
            C.c_float(ext), C.c_uint32(ngmax), _ptr(nidx), _ptr(nc)), "find_neighbors_groups")
        return nidx, nc


_DEFAULT = None


def load(device=0):
    """shared default context (tests)"""
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = Context(device)
    return _DEFAULT
