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
    r = subprocess.run([SEAM_EXE], capture_output=True, text=True, timeout=300)
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
