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
