"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/cstone_hip.h declares.
No compute calls are made here (there is no GPU in the build container)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cstone_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cstone_hip_\w+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    assert len(syms) >= 30
    for must in ("cstone_hip_compute_sfc_keys", "cstone_hip_sort_pairs", "cstone_hip_update_octree",
                 "cstone_hip_build_octree", "cstone_hip_find_halos", "cstone_hip_find_neighbors"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    import cstone_amd

    lib = cstone_amd.load_library()  # raises if the library was not built: there is no fallback
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(cstone_amd.EXPORTS) == [s for s in declared_symbols() if s in cstone_amd.EXPORTS]
    assert set(declared_symbols()) == set(cstone_amd.EXPORTS), set(declared_symbols()) ^ set(cstone_amd.EXPORTS)


def test_fault_injection_is_compiled_into_the_tests_build_only(monkeypatch):
    """the product library is built without -DCSTONE_TEST_HOOKS; lib/libcstone_hip_hooks.so (same sources, the two objects
    with hooks recompiled) is what the failure tests load"""
    import cstone_amd

    assert cstone_amd.load_library().cstone_hip_test_hooks() == 0
    monkeypatch.setenv("CSTONE_HIP_LIB", cstone_amd.LIBPATH.replace("libcstone_hip.so", "libcstone_hip_hooks.so"))
    hooked = cstone_amd.load_library()
    assert hooked.cstone_hip_test_hooks() == 1
    assert not [s for s in declared_symbols() if not hasattr(hooked, s)]


def test_entries_refuse_a_context_that_is_not_alive():
    """cstone_hip_free / _malloc / _ctx_destroy on a pointer that is not a live context (a client that tears down in the
    wrong order, gpurun_out/r3_let1.log) return CSTONE_E_ARG without dereferencing it -- no GPU needed for that"""
    import cstone_amd

    lib = cstone_amd.load_library()
    E_ARG = -1  # CSTONE_E_ARG
    bogus = ctypes.create_string_buffer(512)  # some memory that never was a context
    p = ctypes.cast(bogus, ctypes.c_void_p)
    assert lib.cstone_hip_free(p, None) == E_ARG
    assert lib.cstone_hip_ctx_destroy(p) == E_ARG
    out = ctypes.c_void_p()
    assert lib.cstone_hip_malloc(p, ctypes.byref(out), ctypes.c_size_t(16)) == E_ARG and not out.value
    assert lib.cstone_hip_free(None, None) == E_ARG


def test_box_pod_layout_matches_header():
    import cstone_amd

    assert ctypes.sizeof(cstone_amd.CBox) == 6 * 8 + 4 * 4
    assert cstone_amd.CBox.bc.offset == 48


def test_product_does_not_reference_the_oracle():
    """the shipped sources must never include, link or call anything under oracle/"""
    src = os.path.join(ROOT, "cornerstone-octree_amd")
    for dirpath, _, files in os.walk(src):
        if "build" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".hip", ".hpp", ".h", ".py", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("oracle/", "cstone_oracle", "libcstone_ref", "import oracle", "from oracle",
                               "cstone_ref_"):
                    assert needle not in text, (dirpath, f, needle)


def test_header_is_plain_c_and_struct_layouts_match_the_bindings(tmp_path):
    """include/cstone_hip.h compiles as C99 (the boundary is a C ABI, not a C++ one) and the structs the ctypes bindings
    mirror have the sizes the C compiler gives them"""
    import subprocess

    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include "cstone_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu\\n", sizeof(cstone_box), sizeof(cstone_hip_domain_view),\n'
                   '  sizeof(cstone_hip_domain_mr_view), sizeof(cstone_hip_domain_mr_octree), sizeof(cstone_hip_comm_ops));\n'
                   '  return 0; }\n')
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe)], check=True, capture_output=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    import cstone_amd
    from cstone_amd.distributed import CommOps, MrOctree, MrView
    from cstone_amd.domain import DomainView

    assert sizes == [ctypes.sizeof(cstone_amd.CBox), ctypes.sizeof(DomainView), ctypes.sizeof(MrView),
                     ctypes.sizeof(MrOctree), ctypes.sizeof(CommOps)]
