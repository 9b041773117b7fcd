#!/usr/bin/env python3
"""Generates tests/golden/ref_domain_mpi_P*.npz by running the REFERENCE's cstone::Domain<uint64_t,double,CpuTag> on
several MPI ranks (oracle/_ref/ref_domain_mpi, built by `make -C oracle refdomain`; needs /root/reference and the MPICH
under /opt/conda, i.e. it runs in the build container only).  The fixtures hold inputs and, per sync and rank: the box,
the rank's SFC range, the global tree, keys / x / h of the assigned particles, the locally essential (focus) tree with
its leaf counts, startCell / endCell, layout(), and the halo particles (x, y, z, h, keys in buffer order)."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_domain_mpi")
MPIEXEC = "/opt/conda/bin/mpiexec"


def cloud(n, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "dups":  # three very tight blobs: with 30-bit keys a fifth of the particles share their key with another
        centers = rng.uniform(0.3, 0.7, (3, 3))
        pos = np.where(rng.uniform(size=(n, 1)) < 0.3, rng.uniform(0, 1, (n, 3)),
                       centers[rng.integers(0, 3, n)] + rng.normal(0, 0.004, (n, 3)))
        pos = np.clip(pos, 0.0, 1.0 - 2.0**-20)
        return pos, 0.01 * rng.uniform(0.5, 1.0, n), rng.integers(0, 1 << 30, n)
    if kind == "uniform":
        pos = rng.uniform(0, 1, (n, 3))
    else:  # half uniform background, half in four blobs: an imbalanced decomposition
        centers = rng.uniform(0.2, 0.8, (4, 3))
        pos = np.where(rng.uniform(size=(n, 1)) < 0.5, rng.uniform(0, 1, (n, 3)),
                       centers[rng.integers(0, 4, n)] + rng.normal(0, 0.04, (n, 3)))
    pos = np.clip(pos, 0.0, 1.0 - 2.0**-30)
    h = 0.02 * rng.uniform(0.5, 1.0, n)
    owner = rng.integers(0, 1 << 30, n)
    return pos, h, owner


def run(P, n, syncs, bucket, bucket_focus, bc, kind, seed, key_bits=64, real_bits=64):
    kdt = np.uint64 if key_bits == 64 else np.uint32
    rdt = np.float64 if real_bits == 64 else np.float32
    ks, rs = key_bits // 8, real_bits // 8
    pos, h, owner = cloud(n, seed, kind)
    if real_bits == 32:  # the driver narrows on reading: keep the fixture's inputs exactly what the domain sees
        pos = np.minimum(pos.astype(rdt), rdt(1.0 - 2.0**-20)).astype(np.float64)
        h = h.astype(rdt).astype(np.float64)
    owner = (owner % P).astype(np.int32)
    lim = np.array([0, 1, 0, 1, 0, 1], dtype=np.float64)
    with tempfile.TemporaryDirectory() as tmp:
        inp = os.path.join(tmp, "in.bin")
        with open(inp, "wb") as f:
            f.write(struct.pack("8q", n, P, syncs, bucket, bucket_focus, *bc))
            f.write(lim.tobytes())
            for d in range(3):
                f.write(np.ascontiguousarray(pos[:, d]).tobytes())
            f.write(h.tobytes())
            f.write(owner.tobytes())
        subprocess.run([MPIEXEC, "-n", str(P), EXE, inp, os.path.join(tmp, "out"), f"k{key_bits}f{real_bits}"],
                       check=True, timeout=600)
        out = dict(n=n, P=P, syncs=syncs, bucket=bucket, bucket_focus=bucket_focus, bc=np.array(bc), lim=lim,
                   x=pos[:, 0].astype(rdt), y=pos[:, 1].astype(rdt), z=pos[:, 2].astype(rdt), h=h.astype(rdt),
                   owner=owner, key_bits=key_bits, real_bits=real_bits)
        for r in range(P):
            raw = open(os.path.join(tmp, f"out.rank{r}.bin"), "rb").read()
            off = 0
            for s in range(syncs):
                info = np.frombuffer(raw, np.int64, 5, off); off += 40
                st, en, wh, L, _ = [int(v) for v in info]
                out[f"s{s}_r{r}_info"] = info.copy()
                out[f"s{s}_r{r}_lim"] = np.frombuffer(raw, np.float64, 6, off).copy(); off += 48
                out[f"s{s}_r{r}_range"] = np.frombuffer(raw, kdt, 2, off).copy(); off += 2 * ks
                leaves = np.frombuffer(raw, kdt, L + 1, off).copy(); off += ks * (L + 1)
                off += 4 * (L + (L & 1))
                m = en - st
                out[f"s{s}_r{r}_keys"] = np.frombuffer(raw, kdt, m, off).copy(); off += ks * m
                out[f"s{s}_r{r}_x"] = np.frombuffer(raw, rdt, m, off).copy(); off += rs * m
                out[f"s{s}_r{r}_h"] = np.frombuffer(raw, rdt, m, off).copy(); off += rs * m
                nh = wh - m
                out[f"s{s}_r{r}_halos"] = np.frombuffer(raw, rdt, 3 * nh, off).reshape(3, nh).copy(); off += 3 * rs * nh
                # the locally essential (focus) tree, layout() and the halo particles' keys and smoothing lengths
                finfo = np.frombuffer(raw, np.int64, 3, off); off += 24
                Lf = int(finfo[0])
                out[f"s{s}_r{r}_cells"] = finfo[1:].copy()
                out[f"s{s}_r{r}_focus_leaves"] = np.frombuffer(raw, kdt, Lf + 1, off).copy(); off += ks * (Lf + 1)
                out[f"s{s}_r{r}_focus_counts"] = np.frombuffer(raw, np.uint32, Lf, off).copy(); off += 4 * (Lf + (Lf & 1))
                out[f"s{s}_r{r}_layout"] = np.frombuffer(raw, np.uint32, Lf + 1, off).copy()
                off += 4 * (Lf + 1 + ((Lf + 1) & 1))
                out[f"s{s}_r{r}_halo_keys"] = np.frombuffer(raw, kdt, nh, off).copy(); off += ks * nh
                out[f"s{s}_r{r}_halo_h"] = np.frombuffer(raw, rdt, nh, off).copy(); off += rs * nh
                if r == 0:
                    out[f"s{s}_leaves"] = leaves
            assert off == len(raw)
        # global leaf counts (private to the reference's GlobalAssignment): recount from everybody's assigned keys
        for s in range(syncs):
            allk = np.sort(np.concatenate([out[f"s{s}_r{r}_keys"] for r in range(P)]))
            lv = out[f"s{s}_leaves"]
            out[f"s{s}_counts"] = np.diff(np.searchsorted(allk, lv, side="left")).astype(np.uint32)
            assert allk.size == n
    return out


if __name__ == "__main__":
    if not os.path.exists(EXE):
        sys.exit("build oracle/_ref/ref_domain_mpi first: make -C oracle refdomain")
    cases = {
        "ref_domain_mpi_P2_uniform_open": dict(P=2, n=12000, syncs=3, bucket=64, bucket_focus=8, bc=(0, 0, 0),
                                               kind="uniform", seed=101),
        "ref_domain_mpi_P3_blobs_pbc": dict(P=3, n=15000, syncs=3, bucket=64, bucket_focus=8, bc=(1, 1, 1),
                                            kind="blobs", seed=102),
        "ref_domain_mpi_P4_blobs_open": dict(P=4, n=16000, syncs=3, bucket=96, bucket_focus=16, bc=(0, 0, 0),
                                             kind="blobs", seed=103),
        "ref_domain_mpi_P6_uniform_pbc": dict(P=6, n=9000, syncs=3, bucket=32, bucket_focus=8, bc=(1, 0, 1),
                                              kind="uniform", seed=104),
        "ref_domain_mpi_P8_blobs_open": dict(P=8, n=9600, syncs=3, bucket=32, bucket_focus=8, bc=(0, 0, 0),
                                             kind="blobs", seed=105),
        # the other instantiations of Domain<KeyType, T>
        "ref_domain_mpi_P3_k32_f32_blobs_open": dict(P=3, n=12000, syncs=3, bucket=64, bucket_focus=8, bc=(0, 0, 0),
                                                     kind="blobs", seed=106, key_bits=32, real_bits=32),
        "ref_domain_mpi_P2_k64_f32_uniform_pbc": dict(P=2, n=10000, syncs=3, bucket=64, bucket_focus=16, bc=(1, 1, 1),
                                                      kind="uniform", seed=107, key_bits=64, real_bits=32),
        # many equal keys: the order among them after an exchange is not defined by the reference (MPI_ANY_SOURCE),
        # the test compares such particles as multisets per key
        "ref_domain_mpi_P3_k32_f32_dups": dict(P=3, n=12000, syncs=3, bucket=64, bucket_focus=8, bc=(0, 0, 0),
                                               kind="dups", seed=108, key_bits=32, real_bits=32),
    }
    only = sys.argv[1:]
    cases = {k: v for k, v in cases.items() if not only or k in only}
    for name, kw in cases.items():
        o = run(**kw)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **o)
        print(name, {r: o[f"s2_r{r}_info"].tolist() for r in range(kw["P"])}, "global leaves", o["s2_leaves"].size - 1)
