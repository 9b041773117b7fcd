"""The drop-in boundary proper: the REFERENCE's own cstone::Domain<KeyType, T, GpuTag> (its host headers, compiled where
they lie under /root/reference) linked against libcstone_hip.so through cornerstone-octree_amd/shim/cstone_gpu_hip.cpp --
no cstone_gpu, no CUDA, no Thrust -- must give what the reference's Domain<KeyType, T, CpuTag> gives on the same
particles.  oracle/ref_domain_gpu.cpp is the comparison of the reference's test/integration_mpi/domain_gpu.cpp:117-136
(nParticles, startIndex, endIndex, nParticlesWithHalos, global tree, keys, x, a conserved property) as a plain main(),
extended to the focus tree, its counts, the layout, y/z/h with their halos, exchangeHalos and several syncs with moving
particles; oracle/Makefile builds it into oracle/_ref/ (where /root/reference exists), the binary travels to the GPU box.
Several ranks share the one GPU under mpiexec, like the reference's own GPU+MPI tier."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_domain_gpu")
SHIM = os.path.join(ROOT, "cornerstone-octree_amd", "shim", "cstone_gpu_hip.cpp")
MPIEXEC = "/opt/conda/bin/mpiexec"
REF = "/root/reference/include/cstone"


def _need_exe():
    if not os.path.exists(EXE):
        if os.path.isdir(REF):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refdomain"], check=True)
        if not os.path.exists(EXE):
            pytest.skip("oracle/_ref/ref_domain_gpu not built (no /root/reference here)")


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference headers")
def test_shim_defines_every_seam_symbol_the_reference_domain_needs():
    """CPU: the reference's Domain<GpuTag> translation unit + the shim link against libcstone_hip.so with no undefined
    cstone:: symbol left (the link step of oracle/Makefile is the check; here: the binary exists and resolves)"""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refdomain"], check=True)
    assert os.path.exists(EXE)
    r = subprocess.run(["ldd", "-r", EXE], capture_output=True, text=True)
    undefined = [l for l in (r.stdout + r.stderr).splitlines() if "undefined symbol" in l and "cstone" in l]
    assert not undefined, undefined[:5]
    assert "libcstone_hip.so" in r.stdout


@pytest.mark.gpu
def test_reference_gpu_domain_equals_reference_cpu_domain_one_rank():
    _need_exe()
    # the reference's CPU flavour runs OpenMP over all host cores it sees; the test box hands out a share of them
    env = dict(os.environ, OMP_NUM_THREADS="8", OMP_WAIT_POLICY="passive")
    r = subprocess.run([EXE, "20000", "3"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.count("PASS") == 6 and "FAIL" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(MPIEXEC), reason="no MPI launcher in this image")
@pytest.mark.parametrize("ranks,n", [(2, 5000), (3, 4000), (5, 2000)])
def test_reference_gpu_domain_equals_reference_cpu_domain_mpi(ranks, n):
    """the reference's GPU+MPI tier (domain_gpu on 2 and 5 ranks, test/integration_mpi/CMakeLists.txt:53-59): focus tree
    (LET), peers, treelet exchange, halo layout and exchange all run through the reference's own host code on top of the
    HIP kernels"""
    _need_exe()
    env = dict(os.environ, OMP_NUM_THREADS="2", OMP_WAIT_POLICY="passive")
    # a rank that finds a difference stops syncing while its peers wait in MPI: the timeout ends such a run
    r = subprocess.run([MPIEXEC, "-n", str(ranks), EXE, str(n), "3"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.count("PASS") == 6 and "FAIL" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
