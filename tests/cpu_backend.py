"""TEST INFRASTRUCTURE: CPU stand-in for py_domain.HipBackend built on the oracle, so that the multi-rank
orchestration (tests/py_domain.py) can be rehearsed with gloo on machines without a GPU.  torch CPU tensors
carry the data; keys travel as int64/int32 bit patterns exactly as on the device."""
import numpy as np
import torch

from oracle import oracle as orc
from py_domain import signed_key


def _k(t, kb):  # torch int tensor -> numpy unsigned view
    return t.numpy().view(np.uint64 if kb == 64 else np.uint32)


def _tk(a):  # numpy unsigned -> torch signed tensor
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype.itemsize == 8 else np.int32).copy())


class CpuBackend:
    def __init__(self):
        self.o = orc.Oracle()
        self.device = torch.device("cpu")

    def make_box(self, lim, bc):
        return orc.Box(lim, bc)

    def minmax(self, a):
        return float(a.min()), float(a.max())

    def minmax3(self, x, y, z):
        return [(float(a.min()), float(a.max())) for a in (x, y, z)]

    def merge_positions(self, keys_a, keys_b, offset, kb):
        ka, kbb = _k(keys_a.contiguous(), kb), _k(keys_b.contiguous(), kb)
        pa = np.arange(ka.size) + np.searchsorted(kbb, ka, side="left") + offset
        pb = np.arange(kbb.size) + np.searchsorted(ka, kbb, side="right") + offset
        return torch.from_numpy(pa.astype(np.int32)), torch.from_numpy(pb.astype(np.int32))

    def gather(self, map_, src, dst):
        dst.copy_(src[map_.long()])

    def scatter(self, map_, src, dst):
        dst[map_.long()] = src

    def gather_scatter(self, map_in, map_out, src, dst):
        dst[map_out.long()] = src[map_in.long()]

    def compute_sfc_keys(self, curve, kb, x, y, z, box):
        if x.numel() == 0:
            return torch.zeros(0, dtype=torch.int64 if kb == 64 else torch.int32)
        return _tk(self.o.compute_sfc_keys(curve, kb, x.contiguous().numpy(), y.contiguous().numpy(),
                                           z.contiguous().numpy(), box))

    def iota(self, n):
        return torch.arange(n, dtype=torch.int32)

    def sort_pairs(self, keys, order):
        kb = keys.element_size() * 8
        k, v = self.o.sort_pairs(_k(keys, kb), order.numpy().view(np.uint32))
        keys.copy_(_tk(k))
        order.copy_(torch.from_numpy(v.view(np.int32).copy()))

    def gather_new(self, order, a):
        return a[order.long()].contiguous()

    def zeros_keys(self, n, kb):
        return torch.zeros(n, dtype=torch.int64 if kb == 64 else torch.int32)

    def zeros_i32(self, n):
        return torch.zeros(n, dtype=torch.int32)

    def set_tree(self, tree, counts, leaves, kb, c0):
        tree[:len(leaves)] = torch.tensor([signed_key(k, kb) for k in leaves], dtype=tree.dtype)
        counts[:len(leaves) - 1] = c0

    def update_octree(self, keys, bucket, tree, counts, nl):
        kb = keys.element_size() * 8
        t, c, conv = self.o.update_octree(_k(keys, kb), bucket, _k(tree[:nl + 1], kb).copy(),
                                          counts[:nl].numpy().view(np.uint32).copy())
        n2 = c.size
        if n2 > counts.numel():
            return -n2, False
        tree[:n2 + 1] = _tk(t)
        counts[:n2] = torch.from_numpy(c.view(np.int32).copy())
        return n2, conv

    def compute_octree_buffers(self, keys, bucket, kb):
        t, c = self.o.compute_octree(_k(keys, kb), bucket)
        nl = c.size
        cap = int(nl * 1.5) + 4096
        tb, cb = self.zeros_keys(cap + 1, kb), self.zeros_i32(cap)
        tb[:nl + 1] = _tk(t)
        cb[:nl] = torch.from_numpy(c.view(np.int32).copy())
        return tb, cb, nl

    def compute_node_counts(self, tree, num_leaves, keys, counts):
        kb = tree.element_size() * 8
        c = self.o.node_counts(_k(tree[:num_leaves + 1], kb).copy(), _k(keys.contiguous(), kb))
        counts[:num_leaves] = torch.from_numpy(c.view(np.int32).copy())

    def build_octree(self, tree, num_leaves):
        kb = tree.element_size() * 8
        return self.o.build_octree(_k(tree[:num_leaves + 1], kb).copy())

    def to_numpy(self, t):
        return t.numpy()

    def keys_to_numpy(self, t, kb):
        return _k(t, kb).copy()

    def searchsorted(self, keys, bounds, kb):
        k = _k(keys, kb)
        return [int(np.searchsorted(k, np.array(b, dtype=k.dtype), side="left")) for b in bounds]

    def find_leaf(self, tree, nl, key, kb, below):
        t = _k(tree[:nl + 1], kb)
        key = np.array(key, dtype=t.dtype)
        if below:
            return int(np.searchsorted(t, key, side="right")) - 1
        return int(np.searchsorted(t, key, side="left"))

    def layout_from_counts(self, counts, nl):
        layout = torch.zeros(nl + 1, dtype=torch.int32)
        layout[1:] = torch.cumsum(counts[:nl].long(), 0).to(torch.int32)
        return layout

    def halo_radii(self, h, layout, first, last, nl, ext):
        return torch.from_numpy(self.o.halo_radii(h.contiguous().numpy(), layout.numpy().view(np.uint32).copy(), first,
                                                  last, nl, ext))

    def halo_boxes(self, curve, tree, radii, box, first, last, rb):
        kb = tree.element_size() * 8
        nl = radii.numel()
        return torch.from_numpy(self.o.halo_boxes(curve, _k(tree[:nl + 1], kb).copy(), radii.numpy(), box, first, last,
                                                  rb))

    def find_overlaps(self, curve, octree, tree, boxes, first, last):
        kb = tree.element_size() * 8
        nl = octree["num_leaves"]
        return torch.from_numpy(self.o.find_overlaps(curve, _k(tree[:nl + 1], kb).copy(), boxes.numpy(), first, last))

    def particles_of_flagged(self, flags, layout, first, last):
        f = flags[first:last].bool()
        counts = (layout[first + 1:last + 1] - layout[first:last]).long()
        mask = torch.repeat_interleave(f, counts)
        return (torch.nonzero(mask, as_tuple=False).flatten() + int(layout[first])).to(torch.int32)
