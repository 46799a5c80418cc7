"""CPU tests: the oracle restatement (oracle/rrtmgp_oracle.cpp) against the fixtures generated from the
reference's own kernel text (oracle/make_golden.py). This is what pins the oracle."""
import os
import numpy as np
import pytest

import cases


@pytest.mark.parametrize("path", cases.golden_files("chain_"), ids=os.path.basename)
def test_oracle_chain_matches_reference(path, oracle_f64, oracle_f32):
    G = np.load(path)
    be = oracle_f64 if cases.dtype_of(G) == np.float64 else oracle_f32
    worst = cases.run_chain_case(be, G, tol=1e-13 if be is oracle_f64 else 2e-6)
    assert worst


@pytest.mark.parametrize("path", cases.golden_files("random_"), ids=os.path.basename)
def test_oracle_random_solvers_match_reference(path, oracle_f64, oracle_f32):
    G = np.load(path)
    be = oracle_f64 if cases.dtype_of(G) == np.float64 else oracle_f32
    worst = cases.run_random_case(be, G, tol=1e-13 if be is oracle_f64 else 2e-6)
    assert worst


@pytest.mark.parametrize("path", cases.golden_files("glue_"), ids=os.path.basename)
def test_oracle_incident_flux_convention_vs_reference_text(path, oracle_f64, oracle_f32):
    """SURVEY Q3 pinned: the reference's CUDA text halves a LW incident flux; the restatement (CPU/Fortran semantics) does
    not, so it reproduces the fixture when handed half the flux."""
    G = np.load(path)
    be = oracle_f64 if cases.dtype_of(G) == np.float64 else oracle_f32
    worst = cases.run_glue_case(be, G, tol=1e-13 if be is oracle_f64 else 2e-6)
    assert worst


def test_golden_set_is_complete():
    names = {os.path.basename(p) for p in cases.golden_files("")}
    for tag in ("f64", "f32"):
        for top in (0, 1):
            assert f"chain_{tag}_top{top}.npz" in names and f"random_{tag}_top{top}.npz" in names
            assert f"glue_{tag}_top{top}.npz" in names
