"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  The multi-rank Domain::sync orchestrated from Python over the per-kernel
C ABI (or, in the CPU suite, over the oracle through tests/cpu_backend.py) with torch.distributed as the transport.  It
restates what csrc/domain_mr.hip does inside the library and exists so that the N > 1 control flow (collective order,
assignment rules, exchange bookkeeping) can be rehearsed with gloo on machines without a GPU and compared with the
reference-under-MPI fixtures; the product's multi-rank path is cstone_hip_domain_mr_* (cstone_amd.distributed wraps it).

One process per rank.  Per sync (the reference's MPI call sites are cited as C1..C5 there):
  C1  global bounding box        : all_reduce(MIN) of 6 values
  C2  global cornerstone tree    : every rank rebalances the replicated leaf array with the same rule and all-reduces
                                   the per-leaf counts (R/tree/update_mpi.hpp:71-94)
      SFC assignment             : uniform bins over the global leaf counts, shifts limited to the neighbouring ranges
                                   (R/domain/domaindecomp.hpp:50-172), identical on all ranks
  C3  particle exchange          : ONE all_to_all_single per field over ranges that are contiguous in SFC order
  C4/C5 halo discovery + exchange: owner-side discovery
"""
import os
import time

import numpy as np


def _torch():
    import torch

    return torch


class Comm:
    """thin torch.distributed wrapper; stages through the host when the backend cannot move device tensors (gloo)"""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.stage = dist.get_backend(group) != "nccl"

    def _to(self, t):
        """tensor on the side the backend communicates from: host for gloo, the GPU for RCCL"""
        if self.stage:
            return t.cpu() if t.is_cuda else t
        return t if t.is_cuda else t.cuda()

    def all_reduce_min(self, t):
        w = self._to(t).clone()
        self.dist.all_reduce(w, op=self.dist.ReduceOp.MIN, group=self.group)
        return w.to(t.device)

    def all_reduce_sum_(self, t):
        w = self._to(t)
        self.dist.all_reduce(w, op=self.dist.ReduceOp.SUM, group=self.group)
        if w is not t:
            t.copy_(w)
        return t

    def count_matrix(self, send_counts):
        """m[src][dst] of every rank's send counts: one all_gather instead of an all_to_all plus reductions"""
        torch = _torch()
        s = torch.tensor(send_counts, dtype=torch.int64)
        if not self.stage:
            s = s.cuda()
        out = torch.empty(self.size * self.size, dtype=torch.int64, device=s.device)
        self.dist.all_gather_into_tensor(out, s, group=self.group)
        return out.cpu().view(self.size, self.size).tolist()

    def exchange_counts(self, send_counts):
        torch = _torch()
        s = torch.tensor(send_counts, dtype=torch.int64)
        r = torch.empty_like(s)
        if not self.stage:
            s, r = s.cuda(), r.cuda()
        self.dist.all_to_all_single(r, s, group=self.group)
        return [int(v) for v in r.cpu().tolist()]

    def all_to_all_v(self, send, send_counts, recv_counts):
        """send: 1-D (or [n, k]) tensor ordered by destination rank; returns the received rows ordered by source rank"""
        torch = _torch()
        src = self._to(send).contiguous()
        out = torch.empty((sum(recv_counts),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        self.dist.all_to_all_single(out, src, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                                    group=self.group)
        return out.to(send.device)

    def all_gather_v(self, t):
        """rows of every rank (variable count), as a list indexed by rank"""
        torch = _torch()
        counts = self.exchange_counts([t.shape[0]] * self.size)
        src = self._to(t).contiguous()
        outs = [torch.empty((c,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device) for c in counts]
        if self.stage:
            self.dist.all_gather(outs, src, group=self.group) if len(set(counts)) == 1 else self._gather_uneven(outs, src)
        else:
            self._gather_uneven(outs, src)
        return [o.to(t.device) for o in outs]

    def _gather_uneven(self, outs, src):
        # all_gather needs equal shapes: pad to the maximum row count
        torch = _torch()
        m = max(o.shape[0] for o in outs)
        pad = torch.zeros((m,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        pad[:src.shape[0]] = src
        bufs = [torch.empty_like(pad) for _ in outs]
        self.dist.all_gather(bufs, pad, group=self.group)
        for o, b in zip(outs, bufs):
            o.copy_(b[:o.shape[0]])


def signed_key(v, kb):
    """bit pattern of an unsigned key as the signed integer torch stores"""
    return v - (1 << kb) if v >= 1 << (kb - 1) else v


def log8ceil(n):
    """smallest l with 8^l >= n (R/sfc/common.hpp:134-142)"""
    l = 0
    while 8 ** l < n:
        l += 1
    return l


def initial_domain_splits(num_ranks, level, kb):
    """R/domain/domaindecomp.hpp:242-255: numRanks equal SFC segments, boundaries rounded down to `level` octal digits"""
    end = 1 << (3 * (10 if kb == 32 else 21))
    shift = 3 * ((10 if kb == 32 else 21) - level)
    delta = end // num_ranks
    return [0] + [((i * delta) >> shift) << shift for i in range(1, num_ranks)] + [end]


def spanning_tree(bounds, kb):
    """computeSpanningTree (R/tree/csarray.hpp:508-531): the coarsest cornerstone leaf array that resolves every boundary;
    per segment the canonical cover by maximal aligned power-of-8 nodes (spanSfcRange, R/sfc/common.hpp:376-438)"""
    end = 1 << (3 * (10 if kb == 32 else 21))
    out = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        while a < b:
            size = end
            while size > 1 and (a % size != 0 or size > b - a):
                size //= 8
            out.append(a)
            a += size
    out.append(bounds[-1])
    return out


def uniform_bins(counts, num_bins):
    """R/domain/domaindecomp.hpp:50-75: leaf indices that split the leaf counts into num_bins equal parts"""
    scan = np.zeros(counts.size + 1, dtype=np.uint64)
    np.cumsum(counts.astype(np.uint64), out=scan[1:])
    bin_count = float(scan[-1]) / num_bins
    bins = np.zeros(num_bins + 1, dtype=np.int64)
    bins[-1] = counts.size
    for i in range(1, num_bins):
        target = np.uint64(i * bin_count)
        bins[i] = np.searchsorted(scan, target, side="left")
    return bins


def limit_boundary_shifts(old, new):
    """R/domain/domaindecomp.hpp:140-172: a boundary may move at most into the neighbouring rank's previous range"""
    if old is None or len(old) != len(new):
        return new
    out = new.copy()
    for r in range(1, len(new) - 1):
        out[r] = min(max(int(new[r]), int(old[r - 1])), int(old[r + 1]))
    return out


class DistributedDomain:
    def __init__(self, backend, comm, curve, key_bits, real_bits, bucket, bucket_focus, box_lim, box_bc=(0, 0, 0),
                 halo_ext=1.0):
        self.b, self.c = backend, comm
        self.curve, self.kb, self.rb = curve, key_bits, real_bits
        self.bucket, self.bucket_focus = bucket, bucket_focus
        self.lim = np.array(box_lim, dtype=np.float64)
        self.bc = tuple(int(v) for v in box_bc)
        self.halo_ext = halo_ext
        self.first_call = True
        self.assignment = None  # P+1 boundary keys (python ints)
        self.gtree = self.gcounts = None
        self.g_leaves = 0
        self.ftree = self.fcounts = None
        self.f_leaves = 0
        self.end_key = 1 << (3 * (10 if key_bits == 32 else 21))
        self.stats = {}
        # CSTONE_DIST_TIMING=1: synchronising wall-clock per phase (diagnostics only; it serialises the stream)
        self.timing = {} if os.environ.get("CSTONE_DIST_TIMING") == "1" else None
        self._t0 = None

    def _tick(self, name):
        if self.timing is None:
            return
        torch = _torch()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        now = time.perf_counter()
        if name is not None and self._t0 is not None:
            self.timing[name] = self.timing.get(name, 0.0) + (now - self._t0)
        self._t0 = now

    # ---- C1
    def _update_box(self, x, y, z):
        torch = _torch()
        T = np.float64 if self.rb == 64 else np.float32
        ext = []
        for lo, hi in (self.b.minmax3(x, y, z) if x.numel() else [(float("inf"), float("-inf"))] * 3):
            ext += [lo, -hi]
        g = self.c.all_reduce_min(torch.tensor(ext, dtype=torch.float64)).tolist()
        fit = np.array([g[0], -g[1], g[2], -g[3], g[4], -g[5]], dtype=np.float64)
        for d in range(3):
            if self.bc[d] == 1:
                fit[2 * d], fit[2 * d + 1] = self.lim[2 * d], self.lim[2 * d + 1]
        if self.first_call:
            self.lim = fit
        else:  # limitBoxShrinking, R/sfc/box.hpp:415-431, in T
            for d in range(3):
                lo, hi = T(self.lim[2 * d]), T(self.lim[2 * d + 1])
                ln = T(hi - lo)
                self.lim[2 * d] = min(T(fit[2 * d]), T(lo + T(0.05) * ln))
                self.lim[2 * d + 1] = max(T(fit[2 * d + 1]), T(hi - T(0.05) * ln))
        return self.b.make_box(self.lim, self.bc)

    # ---- C2
    def _update_global_tree(self, keys):
        torch = _torch()
        b = self.b
        if self.gtree is None:
            # GlobalAssignment ctor (R/domain/assignment.hpp:42-53): spanning tree of numRanks equal segments at level
            # log8ceil(100 numRanks), every leaf count = bucketSize - 1
            init = spanning_tree(initial_domain_splits(self.c.size, log8ceil(100 * self.c.size), self.kb), self.kb)
            cap = max(1 << 16, 2 * len(init))
            self.gtree = b.zeros_keys(cap + 1, self.kb)
            self.gcounts = b.zeros_i32(cap)
            b.set_tree(self.gtree, self.gcounts, init, self.kb, self.bucket - 1)
            self.g_leaves = len(init) - 1
        steps = 0
        while True:
            # the decisions use the all-reduced counts of the previous step: identical on every rank
            nl, conv = b.update_octree(keys, self.bucket, self.gtree, self.gcounts, self.g_leaves)
            if nl < 0:  # capacity
                need = -nl + 1
                t2, c2 = b.zeros_keys(2 * need + 1, self.kb), b.zeros_i32(2 * need)
                t2[:self.g_leaves + 1] = self.gtree[:self.g_leaves + 1]
                c2[:self.g_leaves] = self.gcounts[:self.g_leaves]
                self.gtree, self.gcounts = t2, c2
                continue
            self.g_leaves = nl
            self.c.all_reduce_sum_(self.gcounts[:nl])
            steps += 1
            # later calls: exactly one step; first call: one step, then `while (!update)` (assignment.hpp:92-98)
            if not self.first_call or (steps >= 2 and conv):
                return

    def _assign(self):
        counts = self.b.to_numpy(self.gcounts[:self.g_leaves]).view(np.uint32)
        leaves = self.b.keys_to_numpy(self.gtree[:self.g_leaves + 1], self.kb)
        bins = uniform_bins(counts, self.c.size)
        new = [int(leaves[i]) for i in bins]
        new = limit_boundary_shifts(self.assignment, new)
        self.assignment = new
        return new

    # ---- the sync
    def sync(self, x, y, z, h):
        """x,y,z,h: this rank's particles (any order).  Returns dict(keys,x,y,z,h,start,end) with arrays holding
        [halos of lower ranks | assigned particles, SFC sorted | halos of higher ranks].

        Like the reference (R/domain/assignment.hpp:121-127) the fields are NOT reordered before the exchange: only
        the SFC ordering is computed, the leaving particles are picked through it, and everything that stays goes
        from its input position straight to its final slot once the halo counts are known."""
        torch = _torch()
        b, c = self.b, self.c
        rank, P = c.rank, c.size
        self._tick(None)
        box = self._update_box(x, y, z)
        self._tick("box")

        n = x.numel()
        keys = b.compute_sfc_keys(self.curve, self.kb, x, y, z, box)
        order = b.iota(n)
        b.sort_pairs(keys, order)
        self._tick("encode_sort")

        self._update_global_tree(keys)
        bounds = self._assign()
        self._tick("global_tree_assign")

        # C3: particle exchange.  Send ranges are contiguous in the sorted order (createSendRanges)
        cut = b.searchsorted(keys, bounds, self.kb)  # P+1 positions
        send_counts = [cut[p + 1] - cut[p] for p in range(P)]
        dropped = n - cut[P]  # particles flagged for removal sort behind the end of the curve
        matrix = c.count_matrix(send_counts) if P > 1 else [[send_counts[0]]]
        recv_counts = [matrix[p][rank] for p in range(P)]
        moved = sum(send_counts) - send_counts[rank]
        moved_any = sum(sum(row) for row in matrix) - sum(matrix[p][p] for p in range(P))
        kept_keys = keys[cut[rank]:cut[rank + 1]]
        kept_order = order[cut[rank]:cut[rank + 1]]
        na = kept_keys.numel()
        recv, rk = None, None
        if P > 1 and moved_any:
            away_send = [0 if p == rank else send_counts[p] for p in range(P)]
            away_recv = [0 if p == rank else recv_counts[p] for p in range(P)]
            leaving = torch.cat([order[cut[0]:cut[rank]], order[cut[rank + 1]:cut[P]]])
            # x, y, z, h of a particle travel as one row: one collective instead of four
            packed = torch.stack([b.gather_new(leaving, f) for f in (x, y, z, h)], dim=1)
            got = c.all_to_all_v(packed, away_send, away_recv)
            if got.shape[0]:
                recv = [r.contiguous() for r in got.unbind(dim=1)]
                rk = b.compute_sfc_keys(self.curve, self.kb, recv[0], recv[1], recv[2], box)
                ro = b.iota(rk.numel())
                b.sort_pairs(rk, ro)
                recv = [b.gather_new(ro, a) for a in recv]
        nb = rk.numel() if rk is not None else 0
        nm = na + nb
        # positions of the kept (already sorted) and the received (sorted among themselves) particles in their merge
        if nb:
            pos_a, pos_b = b.merge_positions(kept_keys, rk, 0, self.kb)
            keys_m = b.zeros_keys(nm, self.kb)
            b.scatter(pos_a, kept_keys, keys_m)
            b.scatter(pos_b, rk, keys_m)
        else:
            pos_a = pos_b = None
            keys_m = kept_keys

        def place(src, src_recv, dst):
            """field values of the assigned particles -> dst[0:nm] in SFC order"""
            if nb:
                b.gather_scatter(kept_order, pos_a, src, dst)
                b.scatter(pos_b, src_recv, dst)
            else:
                b.gather(kept_order, src, dst)

        h_m = torch.empty(nm, dtype=h.dtype, device=h.device)
        place(h, recv[3] if nb else None, h_m)
        self.stats.update(moved=moved, dropped=dropped, assigned=nm)
        self._tick("particle_exchange_merge")

        # local focus tree (finest resolution inside the assignment)
        if self.ftree is None:
            self.ftree, self.fcounts, self.f_leaves = b.compute_octree_buffers(keys_m, self.bucket_focus, self.kb)
        else:
            self._update_focus(keys_m)
        self._enforce_boundaries(keys_m, (bounds[rank], bounds[rank + 1]))
        L = self.f_leaves
        octree = b.build_octree(self.ftree, num_leaves=L)
        first = b.find_leaf(self.ftree, L, bounds[rank], self.kb, below=True)
        last = b.find_leaf(self.ftree, L, bounds[rank + 1], self.kb, below=False)
        layout = b.layout_from_counts(self.fcounts, L)
        self._tick("focus_tree")

        # C4: owner-side halo discovery: who needs which of my particles
        sel, hs_counts, hr_counts = None, [0] * P, [0] * P
        if P > 1:
            radii = b.halo_radii(h_m, layout[first:], first, last, L, self.halo_ext)
            boxes = b.halo_boxes(self.curve, self.ftree, radii, box, first, last, self.rb)
            mine = boxes[boxes[:, 6] != 0]
            everyone = c.all_gather_v(mine)
            send_idx = []
            for p in range(P):
                if p == rank or everyone[p].shape[0] == 0:
                    continue
                flags = b.find_overlaps(self.curve, octree, self.ftree, everyone[p], first, last)
                idx = b.particles_of_flagged(flags, layout, first, last)
                send_idx.append(idx)
                hs_counts[p] = int(idx.numel())
            hmatrix = c.count_matrix(hs_counts)
            hr_counts = [hmatrix[p][rank] for p in range(P)]
            sel = torch.cat(send_idx) if send_idx else b.iota(0)
            self.stats.update(halos=sum(hr_counts), halo_boxes=int(mine.shape[0]), served=sum(hs_counts))
        nlo, nhi = sum(hr_counts[:rank]), sum(hr_counts[rank + 1:])
        self._tick("halo_discovery")

        # final buffers: [halos of lower ranks | assigned | halos of higher ranks]; every assigned value is written once
        total = nlo + nm + nhi
        out = [torch.empty(total, dtype=f.dtype, device=f.device) for f in (x, y, z, h)]
        for f, fr, o in zip((x, y, z), (recv[:3] if nb else (None,) * 3), out[:3]):
            place(f, fr, o[nlo:nlo + nm])
        out[3][nlo:nlo + nm].copy_(h_m)
        keys_out = b.zeros_keys(total, self.kb)
        keys_out[nlo:nlo + nm].copy_(keys_m)
        self._tick("assemble")

        # C5: halo exchange, one packed collective
        if P > 1:
            packed = torch.stack([b.gather_new(sel, o[nlo:nlo + nm]) for o in out], dim=1)
            got = c.all_to_all_v(packed, hs_counts, hr_counts)
            for d, o in enumerate(out):
                col = got[:, d].contiguous()
                o[:nlo].copy_(col[:nlo])
                o[nlo + nm:].copy_(col[nlo:])
            if nlo:
                keys_out[:nlo].copy_(b.compute_sfc_keys(self.curve, self.kb, out[0][:nlo].contiguous(),
                                                        out[1][:nlo].contiguous(), out[2][:nlo].contiguous(), box))
            if nhi:
                keys_out[nlo + nm:].copy_(b.compute_sfc_keys(self.curve, self.kb, out[0][nlo + nm:].contiguous(),
                                                             out[1][nlo + nm:].contiguous(),
                                                             out[2][nlo + nm:].contiguous(), box))
        self._tick("halo_exchange")
        self.first_call = False
        return dict(keys=keys_out, x=out[0], y=out[1], z=out[2], h=out[3], start=nlo, end=nlo + nm, box=box,
                    lim=self.lim.copy())

    def _enforce_boundaries(self, keys, mandatory):
        """The rank's SFC range must start and end on leaf boundaries of its own tree (the job of enforceKeys in the
        reference's focus tree, R/focus/rebalance.hpp:199-266): a leaf that straddles the range is replaced by the
        coarsest set of octree nodes that resolves the boundary key.  Such leaves are mostly empty, so the count-driven
        update merges them again and the split is redone at every sync -- a copy of the leaf array and a recount."""
        torch = _torch()
        b = self.b
        end = self.end_key
        changed = False
        for key in mandatory:
            if key == 0 or key >= end:
                continue
            L = self.f_leaves
            idx = b.find_leaf(self.ftree, L, key, self.kb, below=True)
            s, e = [int(v) for v in b.keys_to_numpy(self.ftree[idx:idx + 2], self.kb)]
            if s == key:
                continue
            cover = spanning_tree([s, key, e], self.kb)  # starts with s, ends with e
            ins = torch.tensor([signed_key(k, self.kb) for k in cover], dtype=self.ftree.dtype).to(self.ftree.device)
            new = torch.cat([self.ftree[:idx], ins, self.ftree[idx + 2:L + 1]])
            newL = L + len(cover) - 2
            if newL + 1 > self.ftree.numel():
                t2, c2 = b.zeros_keys(2 * newL + 1, self.kb), b.zeros_i32(2 * newL)
                self.ftree, self.fcounts = t2, c2
            self.ftree[:newL + 1] = new
            self.f_leaves = newL
            changed = True
        if changed:
            b.compute_node_counts(self.ftree, self.f_leaves, keys, self.fcounts)

    def _update_focus(self, keys):
        b = self.b
        while True:
            nl, _ = b.update_octree(keys, self.bucket_focus, self.ftree, self.fcounts, self.f_leaves)
            if nl < 0:
                need = -nl + 1
                t2, c2 = b.zeros_keys(2 * need + 1, self.kb), b.zeros_i32(2 * need)
                t2[:self.f_leaves + 1] = self.ftree[:self.f_leaves + 1]
                c2[:self.f_leaves] = self.fcounts[:self.f_leaves]
                self.ftree, self.fcounts = t2, c2
                continue
            self.f_leaves = nl
            return


class HipBackend:
    """adapter from the duck-typed backend interface onto cstone_amd.Context (device tensors)"""

    def __init__(self, ctx):
        import cstone_amd

        self.ctx, self.cs = ctx, cstone_amd

    def make_box(self, lim, bc):
        return self.cs.make_cbox(lim, bc)

    def minmax(self, a):
        return self.ctx.minmax(a)

    def minmax3(self, x, y, z):
        return self.ctx.minmax_arrays([x.contiguous(), y.contiguous(), z.contiguous()])

    def merge_positions(self, keys_a, keys_b, offset, kb):
        return self.ctx.merge_positions(keys_a, keys_b, offset)

    def gather(self, map_, src, dst):
        self.ctx.gather(map_, src, dst)

    def scatter(self, map_, src, dst):
        self.ctx.scatter(map_, src.contiguous(), dst)

    def gather_scatter(self, map_in, map_out, src, dst):
        self.ctx.gather_scatter(map_in, map_out, src, dst)

    def compute_sfc_keys(self, curve, kb, x, y, z, box):
        torch = _torch()
        if x.numel() == 0:
            return torch.zeros(0, dtype=self.cs.key_torch_dtype(kb), device=x.device)
        return self.ctx.compute_sfc_keys(curve, kb, x.contiguous(), y.contiguous(), z.contiguous(), box)

    def iota(self, n):
        torch = _torch()
        t = torch.empty(n, dtype=torch.int32, device=self.ctx.device)
        if n:
            self.ctx.sequence(t)
        return t

    def sort_pairs(self, keys, order):
        if keys.numel():
            self.ctx.sort_pairs(keys, order)

    def gather_new(self, order, a):
        torch = _torch()
        out = torch.empty(order.numel(), dtype=a.dtype, device=a.device)
        if order.numel():
            self.ctx.gather(order, a.contiguous(), out)
        return out

    def zeros_keys(self, n, kb):
        torch = _torch()
        return torch.zeros(n, dtype=self.cs.key_torch_dtype(kb), device=self.ctx.device)

    def zeros_i32(self, n):
        torch = _torch()
        return torch.zeros(n, dtype=torch.int32, device=self.ctx.device)

    def set_tree(self, tree, counts, leaves, kb, c0):
        torch = _torch()
        t = torch.tensor([signed_key(k, kb) for k in leaves], dtype=tree.dtype)
        tree[:len(leaves)] = t.to(tree.device)
        counts[:len(leaves) - 1] = c0

    def update_octree(self, keys, bucket, tree, counts, nl):
        return self.ctx.update_octree(keys, bucket, tree, counts, nl)

    def compute_octree_buffers(self, keys, bucket, kb):
        torch = _torch()
        t, c, it = self.ctx.compute_octree(keys, bucket)
        nl = c.numel()
        cap = int(nl * 1.5) + 4096
        tb, cb = self.zeros_keys(cap + 1, kb), self.zeros_i32(cap)
        tb[:nl + 1] = t
        cb[:nl] = c
        return tb, cb, nl

    def build_octree(self, tree, num_leaves):
        return self.ctx.build_octree(tree, num_leaves=num_leaves)

    def compute_node_counts(self, tree, num_leaves, keys, counts):
        self.ctx.compute_node_counts(tree, keys, counts, num_nodes=num_leaves)

    def to_numpy(self, t):
        return t.cpu().numpy()

    def keys_to_numpy(self, t, kb):
        return self.cs.keys_to_numpy(t, kb)

    def searchsorted(self, keys, bounds, kb):
        torch = _torch()
        q = torch.tensor([signed_key(b, kb) for b in bounds], dtype=keys.dtype, device=keys.device)
        return [int(v) for v in self.ctx.lower_bound(keys, q).cpu().tolist()]

    def find_leaf(self, tree, nl, key, kb, below):
        torch = _torch()
        # leaf START keys only: the end of the 64-bit curve (2^63) is negative in torch's signed view
        if key >= 1 << (3 * (10 if kb == 32 else 21)):
            return nl - 1 if below else nl
        q = torch.tensor([key], dtype=tree.dtype, device=tree.device)
        if below:  # last leaf starting at or before key
            return int(torch.searchsorted(tree[:nl], q, right=True).item()) - 1
        return int(torch.searchsorted(tree[:nl], q, right=False).item())

    def layout_from_counts(self, counts, nl):
        torch = _torch()
        layout = torch.zeros(nl + 1, dtype=torch.int32, device=counts.device)
        if nl:
            self.ctx.inclusive_scan(counts[:nl], layout[1:])
        return layout

    def halo_radii(self, h, layout, first, last, nl, ext):
        return self.ctx.halo_radii(h.contiguous(), layout.contiguous(), first, last, nl, ext)

    def halo_boxes(self, curve, tree, radii, box, first, last, rb):
        return self.ctx.halo_boxes(curve, tree, radii, box, first, last, rb)

    def find_overlaps(self, curve, octree, tree, boxes, first, last):
        return self.ctx.find_overlaps(curve, octree, tree, boxes.contiguous(), first, last)

    def particles_of_flagged(self, flags, layout, first, last):
        torch = _torch()
        f = flags[first:last].bool()
        counts = (layout[first + 1:last + 1] - layout[first:last]).long()
        mask = torch.repeat_interleave(f, counts)
        return (torch.nonzero(mask, as_tuple=False).flatten() + int(layout[first].item())).to(torch.int32)


