"""The C++20 host layer (cornerstone-octree_amd/include/cstone_amd/cstone_amd.hpp) compiles with a plain host compiler
against the C ABI (CPU test) and the example client runs on the GPU (-m gpu)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cornerstone-octree_amd", "build", "domain_example")


SEAM_EXE = os.path.join(ROOT, "cornerstone-octree_amd", "build", "seam_check")


def _stale(exe):
    """the client binary is older than the headers it was compiled against (the ABI structs may have grown since)"""
    if not os.path.exists(exe):
        return True
    heads = [os.path.join(ROOT, "include", "cstone_hip.h"),
             os.path.join(ROOT, "cornerstone-octree_amd", "include", "cstone_amd", "cstone_amd.hpp")]
    return any(os.path.getmtime(h) > os.path.getmtime(exe) for h in heads)


def _check_continuum(stdout):
    """computeContinuumCsarray of the C++ layer (R/tree/continuum.hpp: rebalance and node geometry on the device, the
    concentration on the host) against the reference's own function, built from its headers into oracle/_ref: same
    leaf array, same counts (FNV-1a over both), 32- and 64-bit keys, a constant and a 1/r concentration"""
    import ctypes as C
    import re

    import numpy as np

    ref = os.path.join(ROOT, "oracle", "_ref", "libcstone_ref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/libcstone_ref.so not built")
    lib = C.CDLL(ref)
    if not hasattr(lib, "cstone_ref_continuum"):
        pytest.skip("oracle/_ref predates cstone_ref_continuum")
    lib.cstone_ref_continuum.argtypes = [C.c_int, C.c_int, C.c_double, C.c_uint, C.c_double, C.c_double, C.c_void_p,
                                         C.c_void_p, C.c_int]
    for kb, name in ((64, "u64"), (32, "u32")):
        for kind in (0, 1):
            cap = 1 << 20
            tree = np.zeros(cap + 1, dtype=np.uint64 if kb == 64 else np.uint32)
            counts = np.zeros(cap, dtype=np.uint32)
            L = lib.cstone_ref_continuum(kb, kind, 1e6, 64, -1.0, 1.0, tree.ctypes.data, counts.ctypes.data, cap)
            assert L > 0
            h = 1469598103934665603
            for v in list(tree[:L + 1].astype(np.uint64)) + list(counts[:L].astype(np.uint64)):
                v = int(v)
                for b in range(8):
                    h = ((h ^ ((v >> (8 * b)) & 0xff)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
            m = re.search(rf"continuum {name} kind {kind}: leaves (\d+) particles (\d+) digest ([0-9a-f]+)", stdout)
            assert m, stdout[-1500:]
            assert (int(m.group(1)), int(m.group(2)), int(m.group(3), 16)) == (L, int(counts[:L].sum()), h), (name, kind)


def _compile(source="domain_example.cpp", exe=EXE):
    lib = os.path.join(ROOT, "cornerstone-octree_amd", "lib")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    cmd = ["g++", "-std=c++20", "-O1", "-Wall", "-Wno-comment", "-I", os.path.join(ROOT, "include"), "-I",
           os.path.join(ROOT, "cornerstone-octree_amd", "include"), os.path.join(ROOT, "examples", source),
           "-L", lib, "-lcstone_hip", f"-Wl,-rpath,{lib}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)


def test_cpp_seam_check_compiles():
    """every template of the host layer instantiated for both key widths and both real types (examples/seam_check.cpp)"""
    _compile("seam_check.cpp", SEAM_EXE)
    assert os.path.exists(SEAM_EXE)


@pytest.mark.gpu
def test_cpp_seam_check_runs():
    if not os.path.exists(SEAM_EXE):
        _compile("seam_check.cpp", SEAM_EXE)
    if _stale(SEAM_EXE):
        _compile("seam_check.cpp", SEAM_EXE)
    r = subprocess.run([SEAM_EXE], capture_output=True, text=True, timeout=300)
    _check_continuum(r.stdout)
    assert r.returncode == 0 and "seam check: all passed" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


def test_cpp_layer_compiles_with_host_compiler():
    _compile()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_domain_example_runs():
    if _stale(EXE):
        _compile()
    out = subprocess.run([EXE, "300000"], check=True, capture_output=True, text=True, timeout=120).stdout
    assert "keys sorted: yes" in out
    assert out.count("focus leaves") == 3
    assert "field followed its particles: yes" in out
    assert "counts add up: yes" in out  # Domain::globalTree() / focusTree(), the scratch-tuple signature of sync
    assert "target groups:" in out and "BAD" not in out


MPI_EXE = os.path.join(ROOT, "cornerstone-octree_amd", "build", "domain_mpi_example")
MPI_INC, MPI_LIB, MPIEXEC = "/opt/conda/include", "/opt/conda/lib/libmpi.so.12", "/opt/conda/bin/mpiexec"
have_mpi = os.path.exists(os.path.join(MPI_INC, "mpi.h")) and os.path.exists(MPI_LIB) and os.path.exists(MPIEXEC)


def _compile_mpi():
    lib = os.path.join(ROOT, "cornerstone-octree_amd", "lib")
    os.makedirs(os.path.dirname(MPI_EXE), exist_ok=True)
    cmd = ["g++", "-std=c++20", "-O1", "-Wall", "-Wno-comment", "-I", os.path.join(ROOT, "include"), "-I",
           os.path.join(ROOT, "cornerstone-octree_amd", "include"), "-I", MPI_INC,
           os.path.join(ROOT, "examples", "domain_mpi_example.cpp"), "-L", lib, "-lcstone_hip", f"-Wl,-rpath,{lib}",
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", MPI_LIB, "-Wl,-rpath,/usr/lib/x86_64-linux-gnu",
           "-Wl,-rpath,/opt/conda/lib", "-o", MPI_EXE]
    subprocess.run(cmd, check=True, capture_output=True)


@pytest.mark.skipif(not have_mpi, reason="no MPI in this image")
def test_cpp_mpi_example_compiles():
    """the multi-rank Domain with MPI as the transport behind cstone_hip_comm_ops (examples/domain_mpi_example.cpp)"""
    _compile_mpi()
    assert os.path.exists(MPI_EXE)


@pytest.mark.gpu
@pytest.mark.skipif(not have_mpi, reason="no MPI in this image")
@pytest.mark.parametrize("ranks", [2, 3])
def test_cpp_mpi_example_runs(ranks):
    if _stale(MPI_EXE):
        _compile_mpi()
    r = subprocess.run([MPIEXEC, "-n", str(ranks), MPI_EXE, "100000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all checks passed" in r.stdout and r.stdout.count(": ok") == 3
    assert r.stdout.count("reapplySync ok, exchangeHalos ok") == 2  # the Domain<KeyType, T> class interface on top


SPH_EXE = os.path.join(ROOT, "cornerstone-octree_amd", "build", "sph_density")


def _compile_sph():
    lib = os.path.join(ROOT, "cornerstone-octree_amd", "lib")
    os.makedirs(os.path.dirname(SPH_EXE), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "sph_density.hip"), "-L", lib, "-lcstone_hip", f"-Wl,-rpath,{lib}", "-o", SPH_EXE]
    subprocess.run(cmd, check=True, capture_output=True)


def test_device_header_client_kernel_compiles():
    """include/cstone_hip_device.hpp (cstone_hip::traverseNeighbors for client kernels, the reference's
    traversal/find_neighbors.cuh:436-506) compiles into a client of its own with hipcc for gfx950: examples/sph_density.hip"""
    _compile_sph()
    assert os.path.exists(SPH_EXE)


@pytest.mark.gpu
def test_device_header_density_equals_list_based_sum():
    """the SPH density summed inside the traversal's functor == the same sum over the lists of cstone_hip_find_neighbors,
    bit for bit, and the neighbour counts agree (open, mixed and periodic boxes; f64 and f32)"""
    src = os.path.join(ROOT, "examples", "sph_density.hip")
    hdr = os.path.join(ROOT, "include", "cstone_hip_device.hpp")
    if not os.path.exists(SPH_EXE) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(SPH_EXE):
        _compile_sph()
    r = subprocess.run([SPH_EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "sph_density OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("mismatches 0, list overflows 0") == 3, r.stdout
