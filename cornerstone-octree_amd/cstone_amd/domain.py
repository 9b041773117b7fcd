"""ctypes plumbing for the cstone_hip_domain_* entry points (tests / bench only)."""
import sys
import ctypes as C

import numpy as np

from . import CBox, Context, CstoneError, key_torch_dtype, make_cbox  # noqa: F401


class DomainView(C.Structure):
    _fields_ = [
        ("start_index", C.c_uint32), ("end_index", C.c_uint32), ("num_particles_with_halos", C.c_uint32),
        ("box", CBox),
        ("num_global_leaves", C.c_int32), ("global_leaves", C.c_void_p), ("global_counts", C.c_void_p),
        ("num_focus_leaves", C.c_int32), ("num_focus_nodes", C.c_int32),
        ("focus_leaves", C.c_void_p), ("focus_leaf_counts", C.c_void_p), ("prefixes", C.c_void_p),
        ("child_offsets", C.c_void_p), ("parents", C.c_void_p), ("level_range", C.c_void_p),
        ("internal_to_leaf", C.c_void_p), ("leaf_to_internal", C.c_void_p), ("layout", C.c_void_p),
        ("centers", C.c_void_p), ("sizes", C.c_void_p), ("halo_flags", C.c_void_p), ("sfc_order", C.c_void_p),
        ("halo_radii", C.c_void_p),
        ("expansion_centers", C.c_void_p),
    ]


class DomainStats(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("syncs", "resorts", "resort_fallbacks", "box_redos", "full_sort_fallbacks",
                                          "last_movers")]


class Domain:
    """cstone::Domain<KeyType, T, GpuTag> on one rank; arrays are torch tensors that the call may exchange"""

    def __init__(self, ctx, curve, key_bits, real_bits, bucket, bucket_focus, theta=0.5, box=None, rank=0, nranks=1):
        self.ctx, self.kb, self.rb = ctx, key_bits, real_bits
        self.h = C.c_void_p()
        box = box if box is not None else make_cbox([0, 1] * 3)
        rc = ctx.lib.cstone_hip_domain_create(ctx.h, C.byref(self.h), C.c_int(curve), C.c_int(key_bits),
                                              C.c_int(real_bits), C.c_int(rank), C.c_int(nranks), C.c_uint32(bucket),
                                              C.c_uint32(bucket_focus), C.c_float(theta), C.byref(box))
        ctx._chk(rc, "domain_create")

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.cstone_hip_domain_destroy(self.h)
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

    def sync(self, keys, x, y, z, h, scratch, props=()):
        """returns (keys, x, y, z, h, scratch, props) as tensors after the buffer exchange, truncated to the new size;
        scratch: one tensor, or a list of tensors (cstone_hip_domain_sync_scratch: from three on x, y, z are gathered in
        one pass) -- a list comes back as a list"""
        many = isinstance(scratch, (list, tuple))
        scr = list(scratch) if many else [scratch]
        tensors = [x, y, z, h] + scr + list(props)
        by_ptr = {t.data_ptr(): t for t in tensors}
        n = x.numel()
        pk = C.c_void_p(keys.data_ptr())
        ptrs = [C.c_void_p(t.data_ptr()) for t in (x, y, z, h)]
        sarr = (C.c_void_p * len(scr))(*[t.data_ptr() for t in scr])
        parr = (C.c_void_p * max(1, len(props)))(*[t.data_ptr() for t in props])
        pbytes = (C.c_int * max(1, len(props)))(*[t.element_size() for t in props])
        if many:
            rc = self.ctx.lib.cstone_hip_domain_sync_scratch(self.h, C.byref(pk), C.byref(ptrs[0]), C.byref(ptrs[1]),
                                                             C.byref(ptrs[2]), C.byref(ptrs[3]), C.c_size_t(n), sarr,
                                                             C.c_int(len(scr)), parr, pbytes, C.c_int(len(props)))
        else:
            rc = self.ctx.lib.cstone_hip_domain_sync(self.h, C.byref(pk), C.byref(ptrs[0]), C.byref(ptrs[1]),
                                                     C.byref(ptrs[2]), C.byref(ptrs[3]), C.c_size_t(n), sarr,
                                                     parr, pbytes, C.c_int(len(props)))
        self.ctx._chk(rc, "domain_sync")
        v = self.view()
        m = v.num_particles_with_halos
        out = [by_ptr[p.value] for p in ptrs]
        sout = [by_ptr[sarr[i]] for i in range(len(scr))]
        pout = [by_ptr[parr[i]] for i in range(len(props))]
        return keys[:m], out[0][:m], out[1][:m], out[2][:m], out[3][:m], (sout if many else sout[0]), [t[:m] for t in pout]

    def sync_grav(self, keys, x, y, z, h, m, scratch, props=()):
        """Domain::syncGrav on one rank (cstone_hip_domain_sync_grav): like sync with a LIST of scratch tensors; the masses m
        follow their particles; returns (keys, x, y, z, h, m, scratch, props), view().expansion_centers is set.  Like every
        property, the buffer behind m must offer n elements of the COORDINATES' size (a float32 m next to float64
        coordinates: hand in the first n elements of a tensor of 2 n)"""
        assert all(t.untyped_storage().nbytes() - t.storage_offset() * t.element_size() >= x.numel() * x.element_size()
                   for t in (m, *props)), "property buffers must offer n elements of the coordinates' size"
        import torch

        def whole(t):  # all bytes of t's storage from its first element on
            off = t.storage_offset() * t.element_size()
            return torch.empty(0, dtype=torch.uint8, device=t.device).set_(t.untyped_storage(), off,
                                                                          (t.untyped_storage().nbytes() - off,))

        scr = list(scratch)
        tensors = [x, y, z, h, m] + scr + list(props)
        by_ptr = {t.data_ptr(): t for t in tensors}
        raw = {t.data_ptr(): whole(t) for t in tensors}
        n = x.numel()
        pk = C.c_void_p(keys.data_ptr())
        ptrs = [C.c_void_p(t.data_ptr()) for t in (x, y, z, h, m)]
        sarr = (C.c_void_p * len(scr))(*[t.data_ptr() for t in scr])
        parr = (C.c_void_p * max(1, len(props)))(*[t.data_ptr() for t in props])
        pbytes = (C.c_int * max(1, len(props)))(*[t.element_size() for t in props])
        rc = self.ctx.lib.cstone_hip_domain_sync_grav(self.h, C.byref(pk), C.byref(ptrs[0]), C.byref(ptrs[1]), C.byref(ptrs[2]),
                                                      C.byref(ptrs[3]), C.byref(ptrs[4]), C.c_int(m.element_size() * 8),
                                                      C.c_size_t(n), sarr, C.c_int(len(scr)), parr, pbytes,
                                                      C.c_int(len(props)))
        self.ctx._chk(rc, "domain_sync_grav")
        cnt = self.view().num_particles_with_halos

        def typed(ptr, like):
            # the buffers were exchanged among x, y, z, h, m, scratch and props: a field may now live in a buffer that was
            # handed in with another element type (every buffer offers n elements of the coordinates' size, see the header)
            return raw[ptr][: cnt * like.element_size()].view(like.dtype)

        out = [typed(p.value, t) for p, t in zip(ptrs, (x, y, z, h, m))]
        return (keys[:cnt], *out, [by_ptr[sarr[i]] for i in range(len(scr))],
                [typed(parr[i], props[i]) for i in range(len(props))])

    def update_expansion_centers(self, x, y, z, m):
        """Domain::updateExpansionCenters on arrays laid out like the last sync's results"""
        rc = self.ctx.lib.cstone_hip_domain_update_expansion_centers(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                                                     C.c_void_p(z.data_ptr()), C.c_void_p(m.data_ptr()),
                                                                     C.c_int(m.element_size() * 8))
        self.ctx._chk(rc, "domain_update_expansion_centers")

    def reapply_sync(self, field):
        """Domain::reapplySync: field (n rows of the last sync's input, 1..32 bytes each) in the order of the result"""
        import torch

        row = int(np.prod(field.shape[1:])) if field.dim() > 1 else 1
        v = self.view()
        out = torch.zeros((v.num_particles_with_halos,) + tuple(field.shape[1:]), dtype=field.dtype, device=field.device)
        rc = self.ctx.lib.cstone_hip_domain_reapply_sync(self.h, C.c_void_p(field.data_ptr()), C.c_size_t(field.shape[0]),
                                                         C.c_int(field.element_size() * row), C.c_void_p(out.data_ptr()))
        self.ctx._chk(rc, "domain_reapply_sync")
        return out

    def view(self):
        v = DomainView()
        self.ctx._chk(self.ctx.lib.cstone_hip_domain_view_get(self.h, C.byref(v)), "domain_view_get")
        return v

    SORT_INCREMENTAL, SORT_FROM_SCRATCH, SORT_ALL_DIGITS = 0, 1, 2

    def set_sort_mode(self, mode):
        """how a sync orders the particles (identical results): incremental re-sort (default), radix sort of the digits
        above the previous leaf level, radix sort of all digits (cstone_hip_domain_set_sort_mode)"""
        self.ctx._chk(self.ctx.lib.cstone_hip_domain_set_sort_mode(self.h, C.c_int(mode)), "domain_set_sort_mode")

    def set_speculative_box(self, on):
        self.ctx._chk(self.ctx.lib.cstone_hip_domain_set_speculative_box(self.h, C.c_int(1 if on else 0)),
                      "domain_set_speculative_box")

    def stats(self):
        """counters of the syncs so far (cstone_hip_domain_stats) as a dict"""
        st = DomainStats()
        self.ctx._chk(self.ctx.lib.cstone_hip_domain_stats_get(self.h, C.byref(st)), "domain_stats_get")
        return {k: getattr(st, k) for k, _ in DomainStats._fields_}

    def fetch(self, ptr, count, dtype):
        """device array behind a view pointer -> numpy"""
        a = np.empty(count, dtype=dtype)
        if count:
            rc = self.ctx.lib.cstone_hip_memcpy_d2h(self.ctx.h, a.ctypes.data_as(C.c_void_p), C.c_void_p(ptr),
                                                    C.c_size_t(a.nbytes))
            self.ctx._chk(rc, "memcpy_d2h")
        return a
