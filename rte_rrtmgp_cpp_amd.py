"""Import shim: makes the hyphenated package directory ``rte-rrtmgp-cpp_amd/`` importable as ``rte_rrtmgp_cpp_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rte-rrtmgp-cpp_amd")
_spec = importlib.util.spec_from_file_location(
    "rte_rrtmgp_cpp_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rte_rrtmgp_cpp_amd"] = _mod
_spec.loader.exec_module(_mod)
