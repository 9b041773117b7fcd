"""ctypes plumbing for the multi-rank Domain::sync inside libcstone_hip (csrc/domain_mr.hip, C ABI cstone_hip_domain_mr_*):
the structs of include/cstone_hip.h, the transports behind cstone_hip_comm_ops -- RcclCollectives (RCCL served from C++
inside the library, csrc/comm_rccl.hip) and TorchCollectives (callbacks into torch.distributed, any backend: the gloo
rehearsals of several ranks on one GPU) -- and NativeDistributedDomain, which hands torch tensors to the library and wraps
its result arrays.  Used by tests/ and bench.py only; torch is plumbing here, not part of the product.
"""
import sys
import numpy as np


def _torch():
    import torch

    return torch


import ctypes as C  # noqa: E402

_ALL_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int)
_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_ALL_TO_ALL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t))


class CommOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_reduce", _ALL_REDUCE), ("all_gather", _ALL_GATHER),
                ("all_to_all_v", _ALL_TO_ALL)]


class MrView(C.Structure):
    _fields_ = [("start_index", C.c_uint32), ("end_index", C.c_uint32), ("num_particles_with_halos", C.c_uint32),
                ("pad0_", C.c_uint32), ("box", C.c_byte * 64),
                ("keys", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("h", C.c_void_p),
                ("num_global_leaves", C.c_int32), ("num_focus_leaves", C.c_int32),
                ("global_leaves", C.c_void_p), ("global_counts", C.c_void_p), ("focus_leaves", C.c_void_p),
                ("focus_leaf_counts", C.c_void_p), ("range_start", C.c_uint64), ("range_end", C.c_uint64),
                ("particles_sent", C.c_uint64), ("halos_received", C.c_uint64), ("halos_sent", C.c_uint64),
                ("halo_boxes_exported", C.c_uint64), ("props", C.c_void_p * 16),
                ("resorts", C.c_uint64), ("start_cell", C.c_int32), ("end_cell", C.c_int32),
                ("num_peers", C.c_int32), ("pad1_", C.c_int32), ("layout", C.c_void_p), ("halo_flags", C.c_void_p)]


class MrOctree(C.Structure):
    _fields_ = [("num_leaves", C.c_int32), ("num_nodes", C.c_int32), ("leaves", C.c_void_p),
                ("leaf_counts", C.c_void_p), ("prefixes", C.c_void_p), ("child_offsets", C.c_void_p),
                ("parents", C.c_void_p), ("level_range", C.c_void_p), ("internal_to_leaf", C.c_void_p),
                ("leaf_to_internal", C.c_void_p), ("layout", C.c_void_p), ("centers", C.c_void_p),
                ("sizes", C.c_void_p), ("expansion_centers", C.c_void_p)]


class _DevMem:
    """raw device memory as a __cuda_array_interface__ object (no copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class TorchCollectives:
    """cstone_hip_comm_ops over torch.distributed: RCCL ("nccl") works on the device buffers in place, gloo stages
    through the host"""

    def __init__(self, ctx, group=None):
        import torch.distributed as dist

        self.ctx, self.dist, self.group = ctx, dist, group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self.stage = dist.get_backend(group) != "nccl"
        self.error = None
        self._cbs = (_ALL_REDUCE(self._all_reduce), _ALL_GATHER(self._all_gather), _ALL_TO_ALL(self._all_to_all_v))
        self.ops = CommOps(None, *self._cbs)

    def _tensor(self, ptr, nbytes):
        torch = _torch()
        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8, device=self.ctx.device)
        return torch.as_tensor(_DevMem(ptr, nbytes), device=self.ctx.device)

    def _guard(self, fn):
        try:
            # RCCL collectives of torch.distributed run on torch's CURRENT stream.  The contract of cstone_hip_comm_ops
            # (cstone_hip.h) wants them ordered on the context's stream: that holds by itself while both are the same
            # stream; otherwise the callback waits for the context's stream before and for torch's stream after
            torch = _torch()
            foreign = (not self.stage and
                       getattr(self.ctx, "stream_handle", None) != torch.cuda.current_stream(self.ctx.device).cuda_stream)
            if foreign:
                self.ctx.sync()
            fn()
            if foreign:
                torch.cuda.current_stream(self.ctx.device).synchronize()
            return 0
        except Exception as e:  # surfaces as CSTONE_E_INTERNAL in the library; the text is kept for the caller
            self.error = e
            return 1

    def _all_reduce(self, user, buf, count, dtype, op):
        def run():
            torch = _torch()
            dt = torch.float64 if dtype == 0 else torch.int32
            t = self._tensor(buf, count * (8 if dtype == 0 else 4)).view(dt)
            w = t.cpu() if self.stage else t
            self.dist.all_reduce(w, op=self.dist.ReduceOp.SUM if op == 0 else self.dist.ReduceOp.MIN, group=self.group)
            if self.stage:
                t.copy_(w)
                torch.cuda.synchronize()

        return self._guard(run)

    def _all_gather(self, user, send, recv, nbytes):
        def run():
            torch = _torch()
            s, r = self._tensor(send, nbytes), self._tensor(recv, nbytes * self.size)
            if self.stage:
                sc, rc = s.cpu(), torch.empty(nbytes * self.size, dtype=torch.uint8)
                self.dist.all_gather_into_tensor(rc, sc, group=self.group)
                r.copy_(rc)
                torch.cuda.synchronize()
            else:
                self.dist.all_gather_into_tensor(r, s, group=self.group)

        return self._guard(run)

    def _all_to_all_v(self, user, send, send_bytes, recv, recv_bytes):
        def run():
            torch = _torch()
            sb = [int(send_bytes[p]) for p in range(self.size)]
            rb = [int(recv_bytes[p]) for p in range(self.size)]
            s, r = self._tensor(send, sum(sb)), self._tensor(recv, sum(rb))
            if self.stage:
                sc, rc = s.cpu(), torch.empty(sum(rb), dtype=torch.uint8)
                self.dist.all_to_all_single(rc, sc, output_split_sizes=rb, input_split_sizes=sb, group=self.group)
                r.copy_(rc)
                torch.cuda.synchronize()
            else:
                self.dist.all_to_all_single(r, s, output_split_sizes=rb, input_split_sizes=sb, group=self.group)

        return self._guard(run)


class RcclCollectives:
    """cstone_hip_comm_ops served by RCCL INSIDE libcstone_hip (csrc/comm_rccl.hip): collectives are enqueued on the
    context's stream by C++ code, nothing of Python runs per collective.  torch.distributed (any backend) is only the
    bootstrap channel that carries the 128-byte RCCL id from rank 0 to the other ranks."""

    def __init__(self, ctx, group=None, rank=None, size=None, unique_id=None):
        self.ctx, self.error = ctx, None
        if rank is None:
            import torch.distributed as dist

            rank, size = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.size = rank, size
        if unique_id is None:
            import torch.distributed as dist

            box = [None]
            if rank == 0:
                buf = (C.c_char * 128)()
                ctx._chk(ctx.lib.cstone_hip_comm_rccl_unique_id(ctx.h, buf), "comm_rccl_unique_id")
                box[0] = bytes(buf.raw)
            if size > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            unique_id = box[0]
        self.unique_id = unique_id
        self.h = C.c_void_p()
        ctx._chk(ctx.lib.cstone_hip_comm_rccl_create(ctx.h, C.c_char_p(unique_id), C.c_int(rank), C.c_int(size),
                                                     C.byref(self.h)), "comm_rccl_create")
        self.ops = CommOps()
        ctx._chk(ctx.lib.cstone_hip_comm_rccl_ops(self.h, C.byref(self.ops)), "comm_rccl_ops")

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.cstone_hip_comm_rccl_destroy(self.h)
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


class NativeDistributedDomain:
    """cstone_hip_domain_mr_* (multi-rank Domain::sync inside libcstone_hip) with torch.distributed collectives"""

    HALOS_LET, HALOS_OWNER_SIDE = 0, 1  # CSTONE_MR_HALOS_*

    def __init__(self, ctx, curve, key_bits, real_bits, bucket, bucket_focus, box_lim, box_bc=(0, 0, 0), group=None,
                 coll=None, halo_mode=None, theta=None):
        import cstone_amd

        self.ctx, self.kb, self.rb = ctx, key_bits, real_bits
        # coll: the transport (TorchCollectives: torch.distributed callbacks, any backend; RcclCollectives: RCCL from
        # C++ inside the library)
        self.coll = coll if coll is not None else TorchCollectives(ctx, group)
        self.h = C.c_void_p()
        box = cstone_amd.make_cbox(box_lim, box_bc)
        ctx._chk(ctx.lib.cstone_hip_domain_mr_create(ctx.h, C.byref(self.h), C.c_int(curve), C.c_int(key_bits),
                                                     C.c_int(real_bits), C.c_int(self.coll.rank),
                                                     C.c_int(self.coll.size), C.c_uint32(bucket),
                                                     C.c_uint32(bucket_focus), C.byref(box), C.byref(self.coll.ops)),
                 "domain_mr_create")
        # how halos are found: the reference's locally essential tree (the default) or owner-side discovery
        if halo_mode is not None:
            ctx._chk(ctx.lib.cstone_hip_domain_mr_set_halo_mode(self.h, C.c_int(halo_mode)), "domain_mr_set_halo_mode")
        if theta is not None:
            ctx._chk(ctx.lib.cstone_hip_domain_mr_set_theta(self.h, C.c_float(theta)), "domain_mr_set_theta")
        self._keep = None

    SORT_INCREMENTAL, SORT_FROM_SCRATCH, SORT_ALL_DIGITS = 0, 1, 2  # CSTONE_SORT_*

    def set_sort_mode(self, mode):
        """how a sync orders this rank's particles (identical results): cstone_hip_domain_mr_set_sort_mode"""
        self.ctx._chk(self.ctx.lib.cstone_hip_domain_mr_set_sort_mode(self.h, C.c_int(mode)), "domain_mr_set_sort_mode")

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.cstone_hip_domain_mr_destroy(self.h)
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

    def view(self):
        v = MrView()
        self.ctx._chk(self.ctx.lib.cstone_hip_domain_mr_view_get(self.h, C.byref(v)), "domain_mr_view_get")
        return v

    def sync_grav(self, x, y, z, h, m, props=(), keys=None):
        """Domain::syncGrav: like sync, the masses m follow their particles (they come back as out["m"]) and the focus tree
        is resolved by the vector MAC on the nodes' mass centres (octree()["expansion_centers"] afterwards)"""
        out = self.sync(x, y, z, h, props=props, keys=keys, mass=m)
        return out

    def update_expansion_centers(self, x, y, z, m):
        """Domain::updateExpansionCenters on arrays laid out like the last sync's results"""
        rc = self.ctx.lib.cstone_hip_domain_mr_update_expansion_centers(self.h, C.c_void_p(x.data_ptr()),
                                                                        C.c_void_p(y.data_ptr()), C.c_void_p(z.data_ptr()),
                                                                        C.c_void_p(m.data_ptr()), C.c_int(m.element_size() * 8))
        if rc != 0 and self.coll.error is not None:
            err, self.coll.error = self.coll.error, None
            raise err
        self.ctx._chk(rc, "domain_mr_update_expansion_centers")

    def sync(self, x, y, z, h, props=(), keys=None, mass=None):
        """returns dict(keys, x, y, z, h, start, end[, props]) of tensors that alias the domain-owned result arrays (valid
        until the next but one sync); props: further fields (rows of 1..32 bytes) that follow their particles; keys: optional
        key array whose remove markers flag particles that leave the domain"""
        torch = _torch()
        import cstone_amd

        self._keep = (x, y, z, h, props)  # inputs must outlive the call
        k = len(props)
        parr = (C.c_void_p * max(1, k))(*[t.data_ptr() for t in props])
        rows = [int(np.prod(t.shape[1:])) if t.dim() > 1 else 1 for t in props]
        pbytes = (C.c_int * max(1, k))(*[t.element_size() * r for t, r in zip(props, rows)])
        self._keep = (x, y, z, h, props, keys, mass)
        if mass is not None:
            rc = self.ctx.lib.cstone_hip_domain_mr_sync_grav(
                self.h, C.c_void_p(keys.data_ptr() if keys is not None else 0), C.c_void_p(x.data_ptr()),
                C.c_void_p(y.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(h.data_ptr()), C.c_void_p(mass.data_ptr()),
                C.c_int(mass.element_size() * 8), C.c_size_t(x.numel()), parr, pbytes, C.c_int(k))
        else:
            rc = self.ctx.lib.cstone_hip_domain_mr_sync_keys(self.h, C.c_void_p(keys.data_ptr() if keys is not None else 0),
                                                             C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                                             C.c_void_p(z.data_ptr()), C.c_void_p(h.data_ptr()),
                                                             C.c_size_t(x.numel()), parr, pbytes, C.c_int(k))
        if rc != 0 and self.coll.error is not None:
            err, self.coll.error = self.coll.error, None
            raise err
        self.ctx._chk(rc, "domain_mr_sync")
        v = self.view()
        n = v.num_particles_with_halos
        rdt = torch.float64 if self.rb == 64 else torch.float32
        kdt = cstone_amd.key_torch_dtype(self.kb)
        es = self.rb // 8

        def wrap(ptr, dt, nbytes):
            return torch.as_tensor(_DevMem(ptr, nbytes), device=self.ctx.device).view(dt) if n else \
                torch.empty(0, dtype=dt, device=self.ctx.device)

        out = dict(keys=wrap(v.keys, kdt, n * self.kb // 8), x=wrap(v.x, rdt, n * es), y=wrap(v.y, rdt, n * es),
                   z=wrap(v.z, rdt, n * es), h=wrap(v.h, rdt, n * es), start=v.start_index, end=v.end_index)
        lim = C.cast(C.byref(v.box), C.POINTER(C.c_double))
        out["lim"] = np.array([lim[i] for i in range(6)])
        out["props"] = [wrap(v.props[q], props[q].dtype, n * props[q].element_size() * rows[q])
                        .view((n,) + tuple(props[q].shape[1:])) for q in range(len(props))]
        if mass is not None:  # (the masses travelled as one more property behind the caller's)
            out["m"] = wrap(v.props[len(props)], mass.dtype, n * mass.element_size())
        return out

    def exchange_halos(self, field):
        """Domain::exchangeHalos: field (tensor of num_particles_with_halos rows of 1..32 bytes, laid out like the
        result arrays) gets its halo ranges overwritten with the owners' values"""
        elem = field.element_size() * (int(np.prod(field.shape[1:])) if field.dim() > 1 else 1)
        rc = self.ctx.lib.cstone_hip_domain_mr_exchange_halos(self.h, C.c_void_p(field.data_ptr()), C.c_int(elem))
        if rc != 0 and self.coll.error is not None:
            err, self.coll.error = self.coll.error, None
            raise err
        self.ctx._chk(rc, "domain_mr_exchange_halos")

    def octree(self):
        """Domain::octreeProperties() + layout(): the tree over all local particles (halos included) of the last sync as
        tensors that alias the domain's arrays; the dict is what Context.find_neighbors takes as `octree`"""
        torch = _torch()
        import cstone_amd

        o = MrOctree()
        self.ctx._chk(self.ctx.lib.cstone_hip_domain_mr_octree_get(self.h, C.byref(o)), "domain_mr_octree_get")
        L, M = o.num_leaves, o.num_nodes
        rdt = torch.float64 if self.rb == 64 else torch.float32
        kdt = cstone_amd.key_torch_dtype(self.kb)

        def wrap(ptr, dt, count):
            nbytes = count * torch.empty(0, dtype=dt).element_size()
            return torch.as_tensor(_DevMem(ptr, nbytes), device=self.ctx.device).view(dt)

        return dict(num_leaves=L, num_nodes=M, leaves=wrap(o.leaves, kdt, L + 1),
                    leaf_counts=wrap(o.leaf_counts, torch.int32, L), prefixes=wrap(o.prefixes, kdt, M),
                    child_offsets=wrap(o.child_offsets, torch.int32, M + 1),
                    level_range=wrap(o.level_range, torch.int32, cstone_amd.max_level(self.kb) + 2),
                    internal_to_leaf=wrap(o.internal_to_leaf, torch.int32, M),
                    leaf_to_internal=wrap(o.leaf_to_internal, torch.int32, M), layout=wrap(o.layout, torch.int32, L + 1),
                    centers=wrap(o.centers, rdt, 3 * M).view(M, 3), sizes=wrap(o.sizes, rdt, 3 * M).view(M, 3),
                    expansion_centers=(wrap(o.expansion_centers, rdt, 4 * M).view(M, 4) if o.expansion_centers else None))

    def reapply_sync(self, field):
        """Domain::reapplySync: field (laid out like the INPUT arrays of the last sync, rows of 1..32 bytes) follows its
        particles; returns a tensor laid out like the result arrays whose assigned range is filled"""
        torch = _torch()
        row = int(np.prod(field.shape[1:])) if field.dim() > 1 else 1
        elem = field.element_size() * row
        v = self.view()
        out = torch.zeros((v.num_particles_with_halos,) + tuple(field.shape[1:]), dtype=field.dtype, device=field.device)
        rc = self.ctx.lib.cstone_hip_domain_mr_reapply_sync(self.h, C.c_void_p(field.data_ptr()),
                                                            C.c_size_t(field.shape[0]), C.c_int(elem),
                                                            C.c_void_p(out.data_ptr()))
        if rc != 0 and self.coll.error is not None:
            err, self.coll.error = self.coll.error, None
            raise err
        self.ctx._chk(rc, "domain_mr_reapply_sync")
        return out

    # the accessors the tests use to compare with DistributedDomain / the reference fixtures
    def fetch(self, ptr, count, dtype):
        a = np.empty(count, dtype=dtype)
        if count:
            self.ctx._chk(self.ctx.lib.cstone_hip_memcpy_d2h(self.ctx.h, a.ctypes.data_as(C.c_void_p), C.c_void_p(ptr),
                                                             C.c_size_t(a.nbytes)), "memcpy_d2h")
        return a
