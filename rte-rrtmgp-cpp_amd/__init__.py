"""rte-rrtmgp-cpp_amd: MI355X-native RTE+RRTMGP hot path (HIP kernels behind a C ABI, include/rrx_hip.h).

The directory name contains a hyphen (it mirrors the reference's repo name), so import it through the top-level
shim module ``rte_rrtmgp_cpp_amd`` (rte_rrtmgp_cpp_amd.py at the repo root).

Python here is test/bench plumbing only. ``HipKernels`` raises if the HIP library or a GPU is missing: the
product path has no CPU fallback. The CPU oracle lives in oracle/ and is never imported from this package.
"""
from . import synthetic, pipeline            # noqa: F401
from .hip_kernels import HipKernels, LIB_PATH, load_library   # noqa: F401
