#!/usr/bin/env python3
"""Digests of BASELINE configs[1] (10^7 uniform particles: encode + sort + cornerstone tree): the
REFERENCE's own CPU code (oracle/_ref/libcstone_ref.so) runs on the reference's RandomCoordinates cloud (seed 42, box
[-1, 1]^3, 64-bit Hilbert keys, bucket 64) and the SHA-256 of its outputs is stored; the 240 MB of input are regenerated on
the GPU box by the restated generator (oracle.random_uniform).  Needs /root/reference (run here, not on the GPU box):
    python tests/golden/make_golden_1e7.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    n, bucket, seed = 10_000_000, 64, 42
    box = orc.Box([-1, 1])
    cpu, ref = orc.Oracle(), orc.Reference()
    x, y, z = cpu.random_uniform(n, box, seed)
    keys = ref.compute_sfc_keys(orc.HILBERT, 64, x, y, z, box)
    sk, order = ref.sort_pairs(keys, np.arange(n, dtype=np.uint32))
    leaves, counts = ref.compute_octree(sk, bucket)
    out = dict(n=n, bucket=bucket, seed=seed, box=[-1, 1], curve="hilbert", key_bits=64,
               generator="oracle.random_uniform = std::mt19937(42), x then y then z (test/coord_samples/random.hpp:93-113)",
               producer="oracle/_ref/libcstone_ref.so (the reference's computeSfcKeys, sort_by_key, computeOctree)",
               x_sha256=sha(x), keys_sha256=sha(keys), sorted_keys_sha256=sha(sk), order_sha256=sha(order),
               leaves_sha256=sha(leaves), counts_sha256=sha(counts), num_leaves=int(leaves.size - 1))
    with open(os.path.join(ROOT, "tests", "golden", "ref_1e7_digests.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
