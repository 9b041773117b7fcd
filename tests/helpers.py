"""shared test helpers (host-side only)"""
import numpy as np

from oracle.oracle import Box, end_key, key_dtype, max_level, real_dtype  # noqa: F401


def pad(prefix, length, key_bits):
    """left-align an octal-digit prefix of `length` bits, like the reference's pad() (sfc/common.hpp:91-95)"""
    return prefix << (3 * max_level(key_bits) - length)


class OctreeMaker:
    """builds cornerstone leaf arrays by successive subdivision, mirroring the reference's test utility
    of the same purpose (test/unit/tree/octree_util.hpp): divide() splits the root, divide(a,b,..) splits
    the node reached by child indices a,b,.."""

    def __init__(self, key_bits):
        self.kb = key_bits
        self.nodes = [(0, 0)]  # (start key, level)

    def divide(self, *path):
        key, lvl = 0, 0
        for d in path:
            lvl += 1
            key += d << (3 * (max_level(self.kb) - lvl))
        idx = self.nodes.index((key, lvl))
        step = 1 << (3 * (max_level(self.kb) - lvl - 1))
        self.nodes[idx:idx + 1] = [(key + i * step, lvl + 1) for i in range(8)]
        return self

    def make(self):
        return np.array([k for k, _ in self.nodes] + [end_key(self.kb)], dtype=key_dtype(self.kb))


def random_cloud(n, box, real_bits, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    T = real_dtype(real_bits)
    lim = box.lim.astype(T)
    out = []
    for d in range(3):
        lo, hi = lim[2 * d], lim[2 * d + 1]
        if kind == "uniform":
            v = rng.uniform(lo, hi, n)
        elif kind == "gaussian":
            v = rng.normal((lo + hi) / 2, (hi - lo) / 5, n)
        elif kind == "clustered":
            centers = rng.uniform(lo, hi, 6)
            v = centers[rng.integers(0, 6, n)] + rng.normal(0, (hi - lo) / 60, n)
        else:
            raise ValueError(kind)
        out.append(np.clip(v.astype(T), lo, hi))
    return out


def sorted_cloud(impl, curve, key_bits, n, box, real_bits, seed, kind="uniform"):
    x, y, z = random_cloud(n, box, real_bits, seed, kind)
    keys = impl.compute_sfc_keys(curve, key_bits, x, y, z, box)
    ks, order = impl.sort_pairs(keys, np.arange(n))
    return x[order], y[order], z[order], ks
