#!/usr/bin/env python3
"""Generates tests/golden/ref_*.npz by running the REFERENCE's own CPU code (oracle/_ref, built from
/root/reference/include by oracle/Makefile) on seeded inputs.  Only inputs and expected outputs are stored (data,
no reference source).  Re-run in the build container:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import random_cloud  # noqa: E402
from oracle.oracle import HILBERT, MORTON, Box, Reference, build  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    build("ref")
    R = Reference()
    box = Box([-1.3, 2.1, 0.2, 0.9, -5, 7])
    n = 3000
    for kb in (32, 64):
        for rb in (32, 64):
            d = {"lim": box.lim, "n": n}
            x, y, z = random_cloud(n, box, rb, seed=1000 + kb + rb, kind="gaussian")
            d.update(x=x, y=y, z=z)
            for curve, cn in ((MORTON, "morton"), (HILBERT, "hilbert")):
                keys = R.compute_sfc_keys(curve, kb, x, y, z, box)
                ks, order = R.sort_pairs(keys, np.arange(n))
                d[f"{cn}_keys"] = keys
                d[f"{cn}_order"] = order
                # leaf array after every updateOctree step from the root, bucket 16 (Domain does ONE step per sync)
                tree = np.array([0, 1 << (3 * (10 if kb == 32 else 21))], dtype=keys.dtype)
                counts = np.array([n], dtype=np.uint32)
                for it in range(12):
                    tree, counts, conv = R.update_octree(ks, 16, tree, counts)
                    d[f"{cn}_tree_it{it}"] = tree
                    d[f"{cn}_counts_it{it}"] = counts
                    if conv:
                        d[f"{cn}_iters"] = it + 1
                        break
                o = R.build_octree(tree)
                for k in ("prefixes", "child_offsets", "parents", "level_range", "internal_to_leaf", "leaf_to_internal"):
                    d[f"{cn}_oct_{k}"] = o[k]
                if curve == HILBERT:
                    nl = tree.size - 1
                    radii = (np.random.default_rng(5).uniform(0.01, 0.06, nl)).astype(np.float32)
                    d["halo_radii"] = radii
                    for bcn, bc in (("open", (0, 0, 0)), ("pbc", (1, 1, 1))):
                        pb = Box(box.lim, bc)
                        for (f, l) in ((0, nl // 4), (nl // 4, 3 * nl // 4)):
                            d[f"halo_flags_{bcn}_{f}_{l}"] = R.find_halos(HILBERT, o, tree, radii, pb, f, l, rb)
                    cen, siz = R.node_centers(HILBERT, o["prefixes"], box, rb)
                    d["centers"], d["sizes"] = cen, siz
                    layout = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
                    xs, ys, zs = x[order], y[order], z[order]
                    h = (0.08 * np.random.default_rng(6).uniform(0.5, 1.5, n)).astype(x.dtype)
                    d["h_sorted"] = h
                    for bcn, bc in (("open", (0, 0, 0)), ("pbc", (1, 0, 1))):
                        pb = Box(box.lim, bc)
                        nidx, nc = R.find_neighbors(xs, ys, zs, h, 0, n, pb, o, layout, cen, siz, 32)
                        d[f"nc_{bcn}"] = nc
                        d[f"nidx_{bcn}"] = nidx
            np.savez_compressed(os.path.join(OUT, f"ref_k{kb}_f{rb}.npz"), **d)
            print("wrote", f"ref_k{kb}_f{rb}.npz")


if __name__ == "__main__":
    main()
