#!/usr/bin/env python3
"""Generates tests/golden/ref_domain_*.npz by running the REFERENCE's own cstone::Domain<KeyType,T,CpuTag> on one
MPI rank (oracle/_ref/libcstone_ref_domain.so, built from /root/reference/include by oracle/Makefile) over several
sync calls with moving particles.  Data only.  Re-run in the build container: python tests/golden/make_golden_domain.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.dirname(os.path.abspath(__file__))


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def run(name, n, bucket, bucket_focus, lim, bc, kind, steps, remove_every=0, key_bits=64, real_bits=64):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libcstone_ref_domain.so"))
    lib.cstone_refdom_create_typed.restype = C.c_void_p
    d = C.c_void_p(lib.cstone_refdom_create_typed(C.c_int(key_bits), C.c_int(real_bits), C.c_uint(bucket),
                                                  C.c_uint(bucket_focus), C.c_float(0.5), (C.c_double * 6)(*lim),
                                                  (C.c_int * 3)(*bc)))
    kdt = np.uint64 if key_bits == 64 else np.uint32
    rdt = np.float64 if real_bits == 64 else np.float32
    marker = kdt(1 << (63 if key_bits == 64 else 30))  # remove marker 2^(3 maxLevel) (definitions.h:87-91)
    rng = np.random.default_rng(12345)
    lo, hi = np.array(lim[0::2]), np.array(lim[1::2])
    if kind == "uniform":
        pos = rng.uniform(lo, hi, (n, 3))
    else:
        centers = rng.uniform(lo, hi, (5, 3))
        pos = centers[rng.integers(0, 5, n)] + rng.normal(0, (hi - lo) / 50, (n, 3))
        pos = np.clip(pos, lo, hi)
    h = (0.02 * rng.uniform(0.5, 1.5, n)).astype(rdt)
    vel = rng.normal(0, 0.01, (n, 3)) * (hi - lo)
    out = {"n0": n, "bucket": bucket, "bucket_focus": bucket_focus, "lim": np.array(lim, dtype=np.float64),
           "bc": np.array(bc), "steps": steps, "key_bits": key_bits, "real_bits": real_bits}
    x, y, z = [np.ascontiguousarray(pos[:, i]).astype(rdt) for i in range(3)]
    keys_in = None
    for s in range(steps):
        out[f"in{s}_x"], out[f"in{s}_y"], out[f"in{s}_z"], out[f"in{s}_h"] = x.copy(), y.copy(), z.copy(), h.copy()
        if keys_in is not None:
            out[f"in{s}_keys"] = keys_in.copy()
        lib.cstone_refdom_set(d, C.c_size_t(x.size), p(x), p(y), p(z), p(h), p(keys_in) if keys_in is not None else None)
        lib.cstone_refdom_sync(d)
        info = (C.c_long * 16)()
        lib.cstone_refdom_info(d, info)
        start, end, m, ngl, nfl = info[0], info[1], info[2], info[3], info[4]
        box = np.frombuffer(info, dtype=np.float64)[8:14].copy()
        keys = np.zeros(m, kdt)
        xo, yo, zo, ho = [np.zeros(m, rdt) for _ in range(4)]
        gl = np.zeros(ngl + 1, kdt)
        fl = np.zeros(nfl + 1, kdt)
        fc = np.zeros(nfl, np.uint32)
        layout = np.zeros(nfl + 1, np.uint32)
        lib.cstone_refdom_get(d, p(keys), p(xo), p(yo), p(zo), p(ho), p(gl), p(fl), p(fc), p(layout))
        out[f"out{s}_info"] = np.array([start, end, m, ngl, nfl])
        out[f"out{s}_box"] = box
        for k, v in (("keys", keys), ("x", xo), ("y", yo), ("z", zo), ("h", ho), ("global_leaves", gl),
                     ("focus_leaves", fl), ("focus_counts", fc), ("layout", layout)):
            out[f"out{s}_{k}"] = v
        # move the particles (reflect at the box of the FIRST step so that the fitted box keeps changing a little)
        m = int(m)
        vel = vel[:m] if vel.shape[0] >= m else vel
        x = xo + vel[:m, 0] * rng.uniform(0.5, 1.5, m)
        y = yo + vel[:m, 1] * rng.uniform(0.5, 1.5, m)
        z = zo + vel[:m, 2] * rng.uniform(0.5, 1.5, m)
        if bc[0] == 1:
            x = lo[0] + np.mod(x - lo[0], hi[0] - lo[0])
        if bc[1] == 1:
            y = lo[1] + np.mod(y - lo[1], hi[1] - lo[1])
        if bc[2] == 1:
            z = lo[2] + np.mod(z - lo[2], hi[2] - lo[2])
        x, y, z = [np.ascontiguousarray(v).astype(rdt) for v in (x, y, z)]
        if real_bits == 32:  # float rounding may land a wrapped coordinate on the upper box face: keep it inside
            for v, ax in ((x, 0), (y, 1), (z, 2)):
                if bc[ax] == 1:
                    v[v >= rdt(hi[ax])] = rdt(lo[ax])
        h = ho
        keys_in = np.zeros(m, kdt)
        if remove_every and s >= 1:
            keys_in[::remove_every] = marker
    lib.cstone_refdom_destroy(d)
    np.savez_compressed(os.path.join(OUT, f"ref_domain_{name}.npz"), **out)
    print("wrote", name)


if __name__ == "__main__":
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refdomain"], check=True)
    run("uniform_open", 6000, 64, 8, [0, 1, 0, 1, 0, 1], (0, 0, 0), "uniform", 3)
    run("clustered_pbc", 5000, 200, 16, [-1, 1, -2, 2, 0, 3], (1, 1, 1), "clustered", 3)
    run("remove_mixed", 5000, 64, 64, [0, 1, 0, 1, 0, 1], (0, 1, 0), "clustered", 4, remove_every=97)
    # the other instantiations of Domain<KeyType, T>: 30-bit keys (many duplicates, trees down to the last level), float
    run("k32_f32_clustered_open", 6000, 64, 8, [0, 1, 0, 1, 0, 1], (0, 0, 0), "clustered", 3, key_bits=32, real_bits=32)
    run("k64_f32_uniform_pbc", 5000, 128, 16, [-1, 1, -2, 2, 0, 3], (1, 1, 1), "uniform", 3, key_bits=64, real_bits=32)
    run("k32_f64_remove", 5000, 64, 32, [0, 1, 0, 1, 0, 1], (0, 1, 0), "clustered", 4, remove_every=89, key_bits=32,
        real_bits=64)
