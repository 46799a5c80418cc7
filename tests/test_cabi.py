"""CPU tests of the drop-in boundary: librrx_hip.so loads without a GPU and exports every symbol that
include/rrx_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rrx_hip.h")
LIB = os.path.join(ROOT, "rte-rrtmgp-cpp_amd", "lib", "librrx_hip.so")


def declared_symbols():
    text = open(HEADER).read()
    macro = text[text.index("#define RRX_DECLARE"):text.index("RRX_DECLARE(double")]
    plain = text[:text.index("#define RRX_DECLARE")]
    names = set(re.findall(r"\b(rrx_\w+)\s*\(", plain))
    for base in re.findall(r"\b(rrx_\w+)##SFX", macro):
        names.add(base + "_f64"); names.add(base + "_f32")
    return sorted(names)


def test_header_declares_the_reference_launcher_surface():
    names = declared_symbols()
    # the 5 reference launcher namespaces (include_kernels_cuda/*.h) must all be covered
    for must in ("rrx_lw_solver_noscat_f64", "rrx_sw_solver_2stream_f64", "rrx_lw_secants_array_f64",
                 "rrx_apply_BC_0_f64", "rrx_apply_BC_gpt_f64", "rrx_apply_BC_factor_f64",
                 "rrx_interpolation_f64", "rrx_compute_tau_absorption_f64", "rrx_compute_tau_absorption_set_f64", "rrx_compute_tau_rayleigh_f64",
                 "rrx_combine_abs_and_rayleigh_f64", "rrx_compute_planck_source_f64", "rrx_reorder123x321_f64",
                 "rrx_reorder12x21_f64", "rrx_zero_array_f64",
                 "rrx_increment_1scalar_by_1scalar_f64", "rrx_increment_2stream_by_2stream_f64",
                 "rrx_inc_1scalar_by_1scalar_bybnd_f64", "rrx_inc_2stream_by_2stream_bybnd_f64", "rrx_delta_scale_2str_k_f64",
                 "rrx_sum_broadband_f64", "rrx_net_broadband_precalc_f64", "rrx_sum_byband_f64", "rrx_net_byband_full_f64",
                 "rrx_get_from_subset_f64"):
        assert must in names and must.replace("_f64", "_f32") in names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} not built: run __graft_entry__.build()")
    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, f"declared in include/rrx_hip.h but not exported: {missing}"


def test_rccl_library_exports_every_declared_symbol_and_splits_columns_like_the_python_side():
    """include/rrx_rccl.h (the all-gather below Python): every declared symbol is exported; rrx_column_range is the rule of
    sharding.column_range (no GPU or communicator needed for either)."""
    from rte_rrtmgp_cpp_amd import sharding
    text = open(os.path.join(ROOT, "include", "rrx_rccl.h")).read()
    names = sorted(set(re.findall(r"\b(rrx_\w+)\s*\(", text)))
    assert "rrx_allgather_fluxes_f64" in names and "rrx_comm_create" in names
    lib = ctypes.CDLL(os.path.join(ROOT, "rte-rrtmgp-cpp_amd", "lib", "librrx_rccl.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    s, e = ctypes.c_int(), ctypes.c_int()
    for world, ntot in ((1, 5), (2, 45), (3, 16384), (8, 16384), (8, 16389), (6, 6)):
        for r in range(world):
            lib.rrx_column_range(r, world, ntot, ctypes.byref(s), ctypes.byref(e))
            assert (s.value, e.value) == tuple(sharding.column_range(r, world, ntot))


def test_product_path_fails_loudly_without_gpu():
    import torch
    import numpy as np
    import rte_rrtmgp_cpp_amd as R
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        R.HipKernels(np.float64)


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rte-rrtmgp-cpp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", src, re.M), f
                assert "liboracle" not in src and "libref_" not in src, f
                if "dlopen" in src:
                    # the one dynamic load inside the package: the driver loads the RCCL all-gather library on demand (--ngpus)
                    assert f == "test_rte_rrtmgp_gpu.cpp" and "librrx_rccl.so" in src and src.count("dlopen(") == 1, f
