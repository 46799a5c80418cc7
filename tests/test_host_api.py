"""CPU tests of the C++ host layer: the public headers are self-contained C++17 (g++ -fsyntax-only on a TU that includes
every one of them and instantiates the class API the way the reference's drivers do), and the host library exports the
driver entry point. No GPU calls."""
import ctypes

import pytest
import os
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_headers_compile_and_keep_the_reference_signatures(tmp_path):
    src = tmp_path / "api.cpp"
    src.write_text(textwrap.dedent('''
        #include "types.h"
        #include "Array.h"
        #include "Gas_concs.h"
        #include "Optical_props.h"
        #include "Source_functions.h"
        #include "Gas_optics_rrtmgp.h"
        #include "Cloud_optics.h"
        #include "Rte_lw.h"
        #include "Rte_sw.h"
        #include "Fluxes.h"
        #include "rte_solver_kernels_cuda.h"
        #include "gas_optics_rrtmgp_kernels_cuda.h"
        #include "optical_props_kernels_cuda.h"
        #include "fluxes_kernels_cuda.h"
        #include "subset_kernels_cuda.h"
        #include "Radiation_solver.h"
        #include "Netcdf_interface.h"
        // call shapes taken from the reference's drivers (src_test/Radiation_solver.cu:520-526,568-579,815-820)
        void f(Rte_lw_gpu& lw, Rte_sw_gpu& sw, std::unique_ptr<Optical_props_arry_gpu>& op, Source_func_lw_gpu& src,
               Array_gpu<Float,2>& a2, Array_gpu<Float,1>& a1, Array_gpu<Float,3>& a3, Fluxes_broadband_gpu& fl)
        {
            lw.rte_lw(op, Bool(1), src, a2, Array_gpu<Float,2>(), a3, a3, 1);
            sw.rte_sw(op, Bool(0), a1, a2, a2, a2, Array_gpu<Float,2>(), a3, a3, a3);
            fl.reduce(a3, a3, op, Bool(1));
            fl.reduce(a3, a3, a3, op, Bool(1));
            Subset_kernels_cuda::get_from_subset(10, 5, 3, 1, a2.ptr(), a2.ptr(), a2.ptr(), a2.ptr(), a2.ptr(), a2.ptr());
            Rte_solver_kernels_cuda::apply_BC(1, 1, 1, Bool(1), a3.ptr());
            auto s = a2.subset({{ {1, 2}, {1, 3} }});
            Float v = a2({1, 1}); (void)v; (void)s;
        }
        '''))
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", f"-I{ROOT}/include", f"-I{ROOT}/include_test", str(src)], check=True)
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DRTE_USE_SP", f"-I{ROOT}/include", f"-I{ROOT}/include_test", str(src)], check=True)


def test_host_library_exports_driver_entry():
    lib = os.path.join(ROOT, "rte-rrtmgp-cpp_amd", "lib", "librte_rrtmgp_hip.so")
    assert os.path.exists(lib), "run __graft_entry__.build()"
    assert hasattr(ctypes.CDLL(lib), "rrx_host_main")
    assert hasattr(ctypes.CDLL(lib), "rrx_host_selftest_delta_scale_gzero")
    assert os.path.exists(os.path.join(ROOT, "rte-rrtmgp-cpp_amd", "lib", "test_rte_rrtmgp_gpu"))


def test_host_libraries_export_the_cxx_driver_api():
    """Every symbol include_test/rrx_cxx_driver.h declares is exported by both precisions of the host library (what
    bench.py --driver cxx and a foreign host model bind), and creating a driver without a GPU fails with a message, not a crash."""
    import re
    hdr = open(os.path.join(ROOT, "include_test", "rrx_cxx_driver.h")).read()
    names = sorted(set(re.findall(r"\b(rrx_cxx_driver_\w+)\s*\(", hdr)))
    assert len(names) == 7, names
    for sfx in ("", "_sp"):
        lib = ctypes.CDLL(os.path.join(ROOT, "rte-rrtmgp-cpp_amd", "lib", f"librte_rrtmgp_hip{sfx}.so"))
        for n in names:
            assert hasattr(lib, n), (sfx, n)
    import torch
    if not torch.cuda.is_available():
        lib.rrx_cxx_driver_create.restype = ctypes.c_void_p
        lib.rrx_cxx_driver_error.restype = ctypes.c_char_p
        h = lib.rrx_cxx_driver_create(b"/nonexistent", 0, None, 0, 0)
        assert not h and len(lib.rrx_cxx_driver_error()) > 0


def test_rrxb_roundtrip(tmp_path):
    import numpy as np
    from rte_rrtmgp_cpp_amd import rrxio
    p = str(tmp_path / "t.nc")
    rrxio.write(p, dict(a=2, b=3), {"x": (np.arange(6.).reshape(2, 3), ["a", "b"]), "s": (np.array(4.5), []),
                                     "n": (rrxio.strings(["h2o", "co2"], 8)[:, :3].copy(), ["a", "b"])})
    dims, v = rrxio.read(p)
    assert dims == dict(a=2, b=3) and v["x"][0][1, 2] == 5.0 and float(v["s"][0]) == 4.5
    assert bytes(v["n"][0][0].astype(np.uint8)) == b"h2o"


def _hostlib():
    return ctypes.CDLL(os.path.join(ROOT, "rte-rrtmgp-cpp_amd", "lib", "librte_rrtmgp_hip.so"))


def test_netcdf4_backend_roundtrip_and_real_file(tmp_path):
    """the NetCDF-4 (HDF5) backend of Netcdf_file. (i) write -> read round trip of every data type the drivers'
    files hold; (ii) the one real NetCDF-4 file of the reference tree (data/aerosol_optics.nc, a data fixture under
    tests/golden) against values read independently with h5dump; (iii) NetCDF-4 -> RRXB conversion read back in Python."""
    import numpy as np
    from rte_rrtmgp_cpp_amd import rrxio
    lib = _hostlib()
    assert lib.rrx_host_selftest_netcdf4(str(tmp_path).encode()) == 0
    aer = os.path.join(ROOT, "tests", "golden", "aerosol_optics.nc")
    assert lib.rrx_host_selftest_read_aerosol_file(aer.encode()) == 0
    out = str(tmp_path / "aerosol_optics.rrxb")
    assert lib.rrx_host_netcdf_convert(aer.encode(), out.encode(), b"rrxb") == 0
    dims, v = rrxio.read(out)
    assert dims["band_sw"] == 14 and dims["hydrophilic"] == 7 and dims["relative_humidity"] == 12 and dims["hydrophobic"] == 14
    arr, dn = v["mass_ext_sw_hydrophilic"]
    assert dn == ["hydrophilic", "relative_humidity", "band_sw"] and arr.dtype == np.float32
    assert np.allclose(arr[3, 5, :3], [123.197, 166.189, 433.198], rtol=1e-5)
    # and back: RRXB -> NetCDF-4 -> RRXB reproduces every variable bit for bit
    nc2 = str(tmp_path / "again.nc"); rr2 = str(tmp_path / "again.rrxb")
    assert lib.rrx_host_netcdf_convert(out.encode(), nc2.encode(), b"netcdf4") == 0
    assert lib.rrx_host_netcdf_convert(nc2.encode(), rr2.encode(), b"rrxb") == 0
    _, v2 = rrxio.read(rr2)
    assert set(v2) == set(v)
    for k in v:
        assert v[k][1] == v2[k][1] and np.array_equal(v[k][0], v2[k][0], equal_nan=True), k


@pytest.mark.parametrize("version", [1, 2])
@pytest.mark.parametrize("nrecvars", [0, 1, 2])
def test_classic_netcdf_reader(version, nrecvars, tmp_path):
    """Classic NetCDF (CDF-1 / CDF-2), the pre-HDF5 format some published data sets still use: every external type, fixed and
    record variables (one record variable = unpadded records), attributes skipped; read through Netcdf_file and compared with
    what was written (the file is produced by a few lines of struct.pack in tests/cases.py, not by a library)."""
    import numpy as np
    import cases
    from rte_rrtmgp_cpp_amd import rrxio
    rng = np.random.default_rng(5 + version + 10*nrecvars)
    dims = dict(expt=3, site=5, level=7, string_len=4, one=1)
    v = {"pres": (rng.uniform(1, 1e5, (5, 7)), ["site", "level"]),
         "flag": (rng.integers(-100, 100, (5,), dtype=np.int8), ["site"]),
         "small": (rng.integers(-30000, 30000, (7,), dtype=np.int16), ["level"]),
         "count": (rng.integers(-2**30, 2**30, (5, 7), dtype=np.int32), ["site", "level"]),
         "name": (np.frombuffer(b"h2o co2 o3  n2o ch4 ", dtype="S1").reshape(5, 4), ["site", "string_len"]),
         "scalar": (np.array(1360.85), []),
         "odd": (rng.uniform(0, 1, (1,)).astype(np.float32), ["one"])}
    if nrecvars >= 1:
        v["rld"] = (rng.uniform(0, 500, (3, 5, 7)).astype(np.float32), ["expt", "site", "level"])
    if nrecvars >= 2:
        v["t_sfc"] = (rng.uniform(250, 310, (3, 5)), ["expt", "site"])
    path = str(tmp_path / "classic.nc")
    cases.write_netcdf_classic(path, dims, v, version=version, record_dim="expt" if nrecvars else None, numrecs=3 if nrecvars else 0)
    lib = _hostlib()
    out = str(tmp_path / "classic.rrxb")
    assert lib.rrx_host_netcdf_convert(path.encode(), out.encode(), b"rrxb") == 0
    d2, v2 = rrxio.read(out)
    for k, n in dims.items():
        assert d2[k] == n, k
    assert set(v2) == set(v)
    for k, (arr, dn) in v.items():
        got, gdn = v2[k]
        assert gdn == dn, k
        want = arr.view(np.int8) if arr.dtype.kind == "S" else (arr.astype(np.int32) if arr.dtype == np.int16 else arr)
        assert got.dtype == want.dtype and np.array_equal(got, want), k
    buf = ctypes.create_string_buffer(64)
    assert lib.rrx_host_netcdf_get_attr(path.encode(), b"pres", b"units", buf, 64) == 4 and buf.value == b"1e-6"
    assert lib.rrx_host_netcdf_get_attr(path.encode(), b"", b"title", buf, 64) > 0 and buf.value.startswith(b"written by")
    assert lib.rrx_host_netcdf_get_attr(path.encode(), b"pres", b"nope", buf, 64) == -1


def test_ngpus_launcher_stops_the_survivors_when_a_rank_fails(tmp_path):
    """`test_rte_rrtmgp_gpu --ngpus=N` reaps its ranks in completion order; when one fails, the others (which would
    wait for ever in ncclCommInitRank / ncclAllGather) are terminated and the launcher exits non-zero. The rank executable is
    replaced by a script (RRX_DRIVER_EXE): rank 1 fails at once, the others would sleep for a minute. No GPU involved."""
    import subprocess
    import time
    exe = os.path.join(ROOT, "rte-rrtmgp-cpp_amd", "lib", "test_rte_rrtmgp_gpu")
    script = tmp_path / "rank.sh"
    script.write_text("#!/bin/bash\ntest -n \"$RRX_COMM_FILE\" || exit 9\nif [ \"$RRX_RANK\" = 1 ]; then exit 7; fi\nexec sleep 60\n")
    script.chmod(0o755)
    env = dict(os.environ, RRX_DRIVER_EXE=str(script))
    t0 = time.time()
    r = subprocess.run([exe, "--ngpus=3"], env=env, capture_output=True, text=True, timeout=50)
    assert r.returncode == 7, (r.returncode, r.stderr[-300:])
    assert time.time() - t0 < 20, "the launcher waited for the sleeping ranks"
    assert "stopping the other ranks" in (r.stdout + r.stderr)
    # all ranks fine: exit status 0
    script.write_text("#!/bin/bash\nexit 0\n")
    assert subprocess.run([exe, "--ngpus=3"], env=env, timeout=50).returncode == 0
