"""The C++20 host layer (cornerstone-octree_amd/include/cstone_amd/cstone_amd.hpp) compiles with a plain host compiler
against the C ABI (CPU test) and the example client runs on the GPU (-m gpu)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cornerstone-octree_amd", "build", "domain_example")


def _compile():
    lib = os.path.join(ROOT, "cornerstone-octree_amd", "lib")
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    cmd = ["g++", "-std=c++20", "-O1", "-Wall", "-Wno-comment", "-I", os.path.join(ROOT, "include"), "-I",
           os.path.join(ROOT, "cornerstone-octree_amd", "include"), os.path.join(ROOT, "examples", "domain_example.cpp"),
           "-L", lib, "-lcstone_hip", f"-Wl,-rpath,{lib}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.run(cmd, check=True, capture_output=True)


def test_cpp_layer_compiles_with_host_compiler():
    _compile()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_domain_example_runs():
    if not os.path.exists(EXE):
        _compile()
    out = subprocess.run([EXE, "300000"], check=True, capture_output=True, text=True, timeout=120).stdout
    assert "keys sorted: yes" in out
    assert out.count("focus leaves") == 3
    assert "field followed its particles: yes" in out
    assert "target groups:" in out and "BAD" not in out
