"""The CPU boundary of the reference (include/rrtmgp_kernels.h, the 19 bind(C) kernels) as shipped by this build:
CPU-side checks (no compute on a GPU here) -- the header agrees with the reference's, the product library exports every name,
and the reference's own unmodified CPU classes drive a boundary library correctly (here: the oracle; on the GPU box the
product library, tests/test_gpu_cpu_boundary.py)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import cases
import cpu_boundary

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_HEADER = "/root/reference/include/rrtmgp_kernels.h"


@pytest.mark.parametrize("precision", ["dp", "sp"])
def test_header_agrees_with_the_reference_header(precision, tmp_path):
    """Both headers in ONE translation unit: a declaration that differs in any parameter type is a compile error
    (conflicting declaration of a C function). Build container only: the reference tree does not travel."""
    if not os.path.exists(REF_HEADER):
        pytest.skip("/root/reference is not present on this machine")
    tu = tmp_path / "both.cpp"
    tu.write_text(f'#include "{REF_HEADER}"\n#undef RRTMGP_KERNELS_H\n#include "{ROOT}/include/rrtmgp_kernels.h"\n'
                  "int main() { int n = 1; Float a[1] = {1}; rrtmgp_kernels::zero_array_3D(&n, &n, &n, a); return 0; }\n")
    flags = ["-DRTE_USE_CBOOL"] + (["-DRTE_USE_SP"] if precision == "sp" else [])
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror"] + flags + [str(tu)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_header_declares_the_nineteen_names():
    text = open(os.path.join(ROOT, "include", "rrtmgp_kernels.h")).read()
    for n in cpu_boundary.NAMES:
        assert f'extern "C" void {n}(' in text, n
    assert text.count('extern "C" void ') == 19 == len(cpu_boundary.NAMES)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_product_library_exports_the_nineteen_names(dtype):
    path = cpu_boundary.lib_path(dtype)
    if not os.path.exists(path):
        pytest.fail(f"{path} not built: run __graft_entry__.build()")
    lib = ctypes.CDLL(path)
    missing = [n for n in cpu_boundary.NAMES if not hasattr(lib, n)]
    assert not missing, missing
    # host-side entries need no GPU
    a = np.ones((2, 3, 4), dtype=dtype)
    n = [ctypes.c_int(v) for v in (4, 3, 2)]
    lib.zero_array_3D(ctypes.byref(n[0]), ctypes.byref(n[1]), ctypes.byref(n[2]), ctypes.c_void_p(a.ctypes.data))
    assert not a.any()


def test_product_library_is_not_the_oracle():
    out = subprocess.run(["ldd", cpu_boundary.lib_path()], capture_output=True, text=True).stdout
    assert "librrx_hip.so" in out and "oracle" not in out


@pytest.mark.parametrize("path", cases.golden_files("tall_f64_top0_nlay140") + cases.golden_files("tall_f64_top1_nlay60"), ids=os.path.basename)
def test_reference_cpu_classes_drive_the_oracle_through_the_boundary(path, oracle_built):
    """The reference's unmodified Rte_lw / Rte_sw / Fluxes / Optical_props / Source_functions sources linked against the oracle's
    19 symbols reproduce the reference KERNEL TEXT's fluxes: the restatement's ABI is what the reference's callers expect."""
    if not os.path.exists(cpu_boundary.runner("oracle")):
        pytest.skip("oracle/_ref/ref_rte_oracle is built in the build container only (make -C oracle refrte)")
    worst = cases.run_reference_classes_case("oracle", np.load(path), tol=1e-13, sw_tol=1e-13)
    assert worst
