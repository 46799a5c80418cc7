"""GPU tests of the product's CPU boundary library (lib/librrtmgp_kernels_hip.so: include/rrtmgp_kernels.h served by the HIP
device layer, host arrays in and out): the golden replays of tests/cases.py through the 19 bind(C) names, and the reference's
own unmodified CPU classes (src/Rte_lw.cpp, Rte_sw.cpp, Fluxes.cpp, Optical_props.cpp, Source_functions.cpp) linked against it."""
import os

import numpy as np
import pytest

import cases
import cpu_boundary

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def boundary_f64():
    return cpu_boundary.HipCpuBoundary(np.float64)


@pytest.fixture(scope="module")
def boundary_f32():
    return cpu_boundary.HipCpuBoundary(np.float32)


@pytest.mark.parametrize("path", cases.golden_files("chain_"), ids=os.path.basename)
def test_cpu_boundary_chain_matches_reference_golden(path, boundary_f64, boundary_f32):
    """rrtmgp_interpolation, rrtmgp_compute_tau_absorption, rrtmgp_compute_tau_rayleigh, combine_and_reorder_2str,
    rrtmgp_compute_Planck_source, rte_lw_solver_noscat, rte_sw_solver_2stream (per g-point and broadband), rte_sum_broadband,
    rte_net_broadband_precalc -- host arrays through the HIP kernels and back."""
    G = np.load(path)
    be = boundary_f64 if cases.dtype_of(G) == np.float64 else boundary_f32
    worst = cases.run_chain_case(be, G, tol=1e-10 if be is boundary_f64 else 2e-4)
    print(sorted(worst.items(), key=lambda kv: -kv[1])[:4])


@pytest.mark.parametrize("path", cases.golden_files("random_"), ids=os.path.basename)
def test_cpu_boundary_solvers_and_optical_props_match_reference_golden(path, boundary_f64, boundary_f32):
    """+ rte_increment_*, rte_inc_*_bybnd, rte_delta_scale_2str_k, Jacobians, diffuse SW boundary condition."""
    G = np.load(path)
    be = boundary_f64 if cases.dtype_of(G) == np.float64 else boundary_f32
    worst = cases.run_random_case(be, G, tol=1e-10 if be is boundary_f64 else 1e-3)
    print(sorted(worst.items(), key=lambda kv: -kv[1])[:4])


def test_cpu_boundary_byband_and_transposes(boundary_f64, oracle_f64):
    rng = np.random.default_rng(3)
    ngpt, nlev, ncol = 12, 9, 21
    lims = np.array([[1, 4], [5, 5], [6, 12]], dtype=np.int32)
    fu = rng.uniform(0, 9, (ngpt, nlev, ncol)); fd = rng.uniform(0, 9, (ngpt, nlev, ncol))
    b, o = boundary_f64, oracle_f64
    bu, bd = b.sum_byband(fu, lims), b.sum_byband(fd, lims)
    assert cases.rel_err(bu, o.sum_byband(fu, lims)) <= 1e-13
    assert np.array_equal(b.net_byband(bd, bu), bd - bu)
    a3 = rng.uniform(-1, 1, (3, 4, 5))
    assert np.array_equal(b.reorder123x321(a3), np.ascontiguousarray(a3.transpose(2, 1, 0)))
    z = np.ones((2, 3, 4, 5)); b.lib.call("zero_array_4D", 5, 4, 3, 2, z)
    assert not z.any()


@pytest.mark.parametrize("path", cases.golden_files("tall_"), ids=os.path.basename)
def test_reference_cpu_classes_on_the_hip_cpu_boundary(path):
    """oracle/_ref/ref_rte_hip = the reference's unmodified src/{Rte_lw,Rte_sw,Fluxes,Optical_props,Source_functions}.cpp +
    a file-driven main, linked against librrtmgp_kernels_hip.so in the build container. Rte_lw::rte_lw, Rte_sw::rte_sw,
    Fluxes_broadband::reduce, add_to and delta_scale then run on the MI355X; fluxes against the reference kernel text (tall
    fixtures, 60 / 140 layers) and against the oracle (by-band clouds, broadband mode, two angles, incident fluxes)."""
    if not os.path.exists(cpu_boundary.runner("hip")):
        pytest.skip("oracle/_ref/ref_rte_hip is built in the build container only (make -C oracle refrte)")
    worst = cases.run_reference_classes_case("hip", np.load(path), tol=1e-10, sw_tol=1e-7)
    print(sorted(worst.items(), key=lambda kv: -kv[1])[:4])
