import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    """Compile the CPU oracle (gcc, a few seconds). Only tests may do this."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, stdout=subprocess.DEVNULL)
    return True


@pytest.fixture(scope="session")
def oracle_f64(oracle_built):
    import oracle_py
    return oracle_py.CpuKernels("oracle", np.float64)


@pytest.fixture(scope="session")
def oracle_f32(oracle_built):
    import oracle_py
    return oracle_py.CpuKernels("oracle", np.float32)


@pytest.fixture(scope="session")
def hip_f64():
    import rte_rrtmgp_cpp_amd as R
    return R.HipKernels(np.float64)


@pytest.fixture(scope="session")
def hip_f32():
    import rte_rrtmgp_cpp_amd as R
    return R.HipKernels(np.float32)


def pytest_terminal_summary(terminalreporter):
    """Worst relative error each parity test observed (cases.rel_err), fp64 and fp32 alike: a regression that stays inside its
    tolerance still shows up here."""
    try:
        import cases
    except Exception:
        return
    if not cases.WORST:
        return
    terminalreporter.write_line("")
    terminalreporter.write_line("worst relative error per test (cases.rel_err):")
    for test, err in sorted(cases.WORST.items()):
        terminalreporter.write_line(f"  {err:9.2e}  {test.split('::', 1)[-1]}")
