#!/usr/bin/env python
"""acceptance.py -- the reference's two acceptance runs on real rrtmgp-data, for the HIP build.

    RRTMGP_DATA=/path/to/rrtmgp-data python tools/acceptance.py [--case allsky|rfmip|both] [--work DIR]

What it does is what /root/reference/allsky/{make_links.sh,allsky_init.py,allsky_run.py,allsky_check.py} and
/root/reference/rfmip/{make_links.sh,rfmip_init.py,rfmip_run.py,check_rfmip.sh} do around the reference's executable, around
this build's driver (librte_rrtmgp_hip.so:rrx_host_main = test_rte_rrtmgp_gpu) instead:

  all-sky  24 columns x 72 layers (RCE-like analytic profile, 2 of every 3 columns cloudy), `--cloud-optics`; lw_flux_{up,dn} and
           sw_flux_{up,dn,dir} against examples/all-sky/reference/rrtmgp-allsky-{lw,sw}-no-aerosols.nc, failure threshold 1e-5 W m-2
  rfmip    the RFMIP clear-sky experiments x sites of examples/rfmip-clear-sky/inputs/multiple_input4MIPs_...nc; rld, rlu, rsd, rsu
           against examples/rfmip-clear-sky/reference/r??_Efx_RTE-RRTMGP-181204_rad-irf_r1i1p1f1_gn.nc, threshold 5.8e-2 W m-2

The coefficient files (rrtmgp-gas-lw-g256.nc, rrtmgp-gas-sw-g224.nc, rrtmgp-clouds-{lw,sw}.nc) are read as they are: NetCDF-4
through the HDF5 backend of include_test/Netcdf_hdf5.h. rrtmgp-data is not part of the reference tree and not in the build
image, so on this machine the script can only be exercised on a stand-in data tree (tests/test_gpu_host_classes.py does that,
with the CPU oracle providing the "reference" fluxes); with the real data it is the route from "parity unpinned" to the
reference's own pins. Exit status 0 = every variable within its threshold.
"""
import argparse
import ctypes
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rte_rrtmgp_cpp_amd import rrxio      # noqa: E402

HOSTLIB = os.path.join(ROOT, "rte-rrtmgp-cpp_amd", "lib", "librte_rrtmgp_hip.so")
GAS_LW, GAS_SW = "rrtmgp-gas-lw-g256.nc", "rrtmgp-gas-sw-g224.nc"
CLD_LW, CLD_SW = "rrtmgp-clouds-lw.nc", "rrtmgp-clouds-sw.nc"
RFMIP_IN = os.path.join("examples", "rfmip-clear-sky", "inputs", "multiple_input4MIPs_radiation_RFMIP_UColorado-RFMIP-1-2_none.nc")
RFMIP_REF = os.path.join("examples", "rfmip-clear-sky", "reference")
ALLSKY_REF = os.path.join("examples", "all-sky", "reference")
RFMIP_GASES = (("co2", "carbon_dioxide_GM"), ("n2o", "nitrous_oxide_GM"), ("co", "carbon_monoxide_GM"), ("ch4", "methane_GM"),
               ("o2", "oxygen_GM"), ("n2", "nitrogen_GM"), ("ccl4", "carbon_tetrachloride_GM"), ("cfc11", "cfc11_GM"),
               ("cfc12", "cfc12_GM"), ("cfc22", "hcfc22_GM"), ("hfc143a", "hfc143a_GM"), ("hfc125", "hfc125_GM"),
               ("hfc23", "hfc23_GM"), ("hfc32", "hfc32_GM"), ("hfc134a", "hfc134a_GM"), ("cf4", "cf4_GM"))

_lib = None


def hostlib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(HOSTLIB)
    return _lib


def nc_read(path, scratch):
    """dims, {name: (array, dim names)} of a NetCDF-4 or RRXB file."""
    with open(path, "rb") as f:
        magic = f.read(4)
    if magic == b"\x89HDF" or magic[:3] == b"CDF":        # NetCDF-4 or classic NetCDF: through the host library's readers
        tmp = os.path.join(scratch, "_read_%d.rrxb" % (abs(hash(path)) % 10**9))
        if hostlib().rrx_host_netcdf_convert(path.encode(), tmp.encode(), b"rrxb") != 0:
            raise RuntimeError("cannot read " + path)
        out = rrxio.read(tmp)
        os.remove(tmp)
        return out
    return rrxio.read(path)


def nc_attr(path, var, attr):
    buf = ctypes.create_string_buffer(512)
    n = hostlib().rrx_host_netcdf_get_attr(path.encode(), var.encode(), attr.encode(), buf, 512)
    return buf.value.decode() if n >= 0 else None


def run_driver(workdir, *flags):
    argv = [b"test_rte_rrtmgp_gpu"] + [f.encode() for f in flags]
    arr = (ctypes.c_char_p * len(argv))(*argv)
    old = os.getcwd()
    try:
        os.chdir(workdir)
        return hostlib().rrx_host_main(len(argv), arr)
    finally:
        os.chdir(old)


def link_coefficients(data, work):
    os.makedirs(work, exist_ok=True)
    for src, dst in ((GAS_LW, "coefficients_lw.nc"), (GAS_SW, "coefficients_sw.nc"), (CLD_LW, "cloud_coefficients_lw.nc"),
                     (CLD_SW, "cloud_coefficients_sw.nc")):
        p = os.path.join(work, dst)
        if os.path.lexists(p):
            os.remove(p)
        os.symlink(os.path.abspath(os.path.join(data, src)), p)


def band_counts(data, scratch):
    return tuple(nc_read(os.path.join(data, f), scratch)[0]["bnd"] for f in (GAS_LW, GAS_SW))


# ---------------------------------------------------------------------------------------------------------------- all-sky
def allsky_input(n_bnd_lw, n_bnd_sw, n_col=24, n_lay=72):
    """The all-sky example problem (definition: /root/reference/allsky/allsky_init.py:18-176): a 15 km troposphere and a
    stratosphere to 70 km with n_lay/2 layers each, RCE-like moisture and temperature, ozone from the RCEMIP fit, liquid
    cloud above 263 K and ice cloud below 273 K between 100 and 900 hPa in two of every three columns."""
    z_top, z_trop = 70.e3, 15.e3
    half = n_lay // 2
    i = np.arange(1, half + 1)
    zh = np.zeros(n_lay + 1)
    zh[1:half+1] = 2.*i*z_trop/n_lay
    zh[half+1:] = z_trop + 2.*i*(z_top - z_trop)/n_lay
    z = 0.5*(zh[1:] + zh[:-1])

    def profile(zz):
        q_0, z_q1, z_q2, q_t = 0.01864, 4.0e3, 7.5e3, 1.e-8
        T_0, gamma, g, Rd, p0 = 300., 6.7e-3, 9.79764, 287.04, 101480.
        above = zz > z_trop
        q = np.where(above, q_t, q_0*np.exp(-zz/z_q1)*np.exp(-(zz/z_q2)**2))
        T = np.where(above, T_0 - gamma*z_trop/(1. + 0.608*q_0), T_0 - gamma*zz/(1. + 0.608*q))
        Tv, Tv_0 = T*(1. + 0.608*q), (1. + 0.608*q_0)*T_0
        p = p0*(Tv/Tv_0)**(g/(Rd*gamma))
        p = np.where(above, p*np.exp(-(g*(zz - z_trop))/(Rd*Tv)), p)
        return p, q, T

    p_lay, q, T_lay = profile(z)
    p_lev, _, T_lev = profile(zh)
    p_hpa = p_lay/100.
    o3 = np.maximum(1e-13, 3.6478*p_hpa**0.83209*np.exp(-p_hpa/11.3515)*1e-6)
    tile = lambda a: np.ascontiguousarray(np.tile(a[:, None, None], (1, 1, n_col)))
    dims = dict(x=n_col, y=1, lay=n_lay, lev=n_lay+1, band_lw=n_bnd_lw, band_sw=n_bnd_sw)
    L, V = ["lay", "y", "x"], ["lev", "y", "x"]
    v = {"z_lay": (z, ["lay"]), "z_lev": (zh, ["lev"]),
         "p_lay": (tile(p_lay), L), "p_lev": (tile(p_lev), V), "t_lay": (tile(T_lay), L), "t_lev": (tile(T_lev), V),
         "vmr_o3": (tile(o3), L), "vmr_h2o": (tile(q), L)}
    for gas, val in (("co2", 348.e-6), ("ch4", 1650.e-9), ("n2o", 306.e-9), ("n2", 0.7808), ("o2", 0.2095)):
        v["vmr_" + gas] = (np.array(val), [])
    v["emis_sfc"] = (np.full((1, n_col, n_bnd_lw), 0.98), ["y", "x", "band_lw"])
    v["t_sfc"] = (np.full((1, n_col), 300.), ["y", "x"])
    v["mu0"] = (np.full((1, n_col), 0.86), ["y", "x"])
    v["sfc_alb_dir"] = (np.full((1, n_col, n_bnd_sw), 0.06), ["y", "x", "band_sw"])
    v["sfc_alb_dif"] = (np.full((1, n_col, n_bnd_sw), 0.06), ["y", "x", "band_sw"])
    cloudy = (np.arange(1, n_col+1) % 3 > 0)[None, None, :]
    pl, tl = v["p_lay"][0], v["t_lay"][0]
    mask = (pl > 1.e4) & (pl < 9.e4) & cloudy
    lwp = np.where(mask & (tl > 263.), 10., 0.)
    iwp = np.where(mask & (tl < 273.), 10., 0.)
    v["lwp"] = (lwp, L); v["iwp"] = (iwp, L)
    v["rel"] = (np.where(lwp > 0., 0.5*(2.5 + 21.5), 0.), L)
    v["dei"] = (np.where(iwp > 0., 0.5*(10. + 180.), 0.), L)
    return dims, v


def _flux(out_vars, name):
    a = out_vars[name][0]
    return a.reshape(a.shape[0], -1)              # (lev, y, x) -> (lev, col)


def _against(ref_arr, tst, name):
    r = np.asarray(ref_arr, dtype=np.float64)
    r = r.reshape(r.shape[0], -1) if r.ndim > 2 else r
    if r.shape != tst.shape and r.T.shape == tst.shape:
        r = r.T
    if r.shape != tst.shape:
        raise RuntimeError(f"{name}: reference shape {r.shape} vs test shape {tst.shape}")
    if not np.all(np.isfinite(r)) or not np.all(np.isfinite(tst)):
        raise RuntimeError(f"{name}: missing values")
    return float(np.abs(tst - r).max())


def case_allsky(data, work, threshold=1.e-5):
    os.makedirs(work, exist_ok=True)
    n_bnd_lw, n_bnd_sw = band_counts(data, work)
    link_coefficients(data, work)
    dims, v = allsky_input(n_bnd_lw, n_bnd_sw)
    rrxio.write(os.path.join(work, "rte_rrtmgp_input.nc"), dims, v)
    if run_driver(work, "--cloud-optics") != 0:
        raise RuntimeError("driver failed on the all-sky problem")
    _, out = nc_read(os.path.join(work, "rte_rrtmgp_output.nc"), work)
    results = []
    for ref_file, names in (("rrtmgp-allsky-lw-no-aerosols.nc", (("lw_flux_up", "lw_flux_up"), ("lw_flux_dn", "lw_flux_dn"))),
                            ("rrtmgp-allsky-sw-no-aerosols.nc", (("sw_flux_up", "sw_flux_up"), ("sw_flux_dn", "sw_flux_dn"),
                                                                 ("sw_flux_dir", "sw_flux_dn_dir")))):
        _, ref = nc_read(os.path.join(data, ALLSKY_REF, ref_file), work)
        for ref_name, out_name in names:
            d = _against(ref[ref_name][0], _flux(out, out_name), ref_name)
            results.append(("allsky " + ref_name, d, threshold))
    return results


# ------------------------------------------------------------------------------------------------------------------ rfmip
def rfmip_inputs(data, scratch, n_bnd_lw, n_bnd_sw):
    """One rte_rrtmgp_input per RFMIP experiment (field mapping: /root/reference/rfmip/rfmip_init.py:17-108)."""
    path = os.path.join(data, RFMIP_IN)
    dims, v = nc_read(path, scratch)
    n_expt, n_site, n_lay, n_lev = dims["expt"], dims["site"], dims["layer"], dims["level"]
    scale = lambda name: float(nc_attr(path, name, "units") or 1.)
    col = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64).T[:, None, :])        # (site, k) -> (k, 1, site)
    p_min = np.nextafter(1.005183574463, 1e8)          # the k-distribution's lowest reference pressure
    for expt in range(n_expt):
        d = dict(x=n_site, y=1, lay=n_lay, lev=n_lev, band_lw=n_bnd_lw, band_sw=n_bnd_sw)
        L, V = ["lay", "y", "x"], ["lev", "y", "x"]
        o = {"p_lay": (col(v["pres_layer"][0]), L), "p_lev": (np.maximum(col(v["pres_level"][0]), p_min), V),
             "t_lay": (col(v["temp_layer"][0][expt]), L), "t_lev": (col(v["temp_level"][0][expt]), V)}
        emis = np.asarray(v["surface_emissivity"][0], dtype=np.float64)
        alb = np.asarray(v["surface_albedo"][0], dtype=np.float64)
        o["emis_sfc"] = (np.ascontiguousarray(np.tile(emis[None, :, None], (1, 1, n_bnd_lw))), ["y", "x", "band_lw"])
        o["t_sfc"] = (np.asarray(v["surface_temperature"][0][expt], dtype=np.float64)[None, :].copy(), ["y", "x"])
        o["sfc_alb_dir"] = (np.ascontiguousarray(np.tile(alb[None, :, None], (1, 1, n_bnd_sw))), ["y", "x", "band_sw"])
        o["sfc_alb_dif"] = (o["sfc_alb_dir"][0].copy(), ["y", "x", "band_sw"])
        sza = np.asarray(v["solar_zenith_angle"][0], dtype=np.float64)
        o["mu0"] = (np.maximum(0., np.cos(np.deg2rad(sza)))[None, :].copy(), ["y", "x"])
        o["tsi"] = (np.asarray(v["total_solar_irradiance"][0], dtype=np.float64)[None, :].copy(), ["y", "x"])
        o["vmr_h2o"] = (col(v["water_vapor"][0][expt]) * scale("water_vapor"), L)
        o["vmr_o3"] = (col(v["ozone"][0][expt]) * scale("ozone"), L)
        for gas, name in RFMIP_GASES:
            if name in v:
                o["vmr_" + gas] = (np.array(float(v[name][0][expt]) * scale(name)), [])
        yield expt, d, o


def case_rfmip(data, work, threshold=5.8e-2):
    os.makedirs(work, exist_ok=True)
    n_bnd_lw, n_bnd_sw = band_counts(data, work)
    link_coefficients(data, work)
    got = {}
    for expt, dims, v in rfmip_inputs(data, work, n_bnd_lw, n_bnd_sw):
        rrxio.write(os.path.join(work, "rte_rrtmgp_input.nc"), dims, v)
        if run_driver(work) != 0:
            raise RuntimeError(f"driver failed on RFMIP experiment {expt}")
        _, out = nc_read(os.path.join(work, "rte_rrtmgp_output.nc"), work)
        shutil.copyfile(os.path.join(work, "rte_rrtmgp_output.nc"), os.path.join(work, "rte_rrtmgp_output_expt_%02d.nc" % expt))
        for rf, name in (("rld", "lw_flux_dn"), ("rlu", "lw_flux_up"), ("rsd", "sw_flux_dn"), ("rsu", "sw_flux_up")):
            got.setdefault(rf, []).append(_flux(out, name).T)              # (site, level), as the rXX files hold it
    results = []
    for rf, arrs in got.items():
        _, ref = nc_read(os.path.join(data, RFMIP_REF, rf + "_Efx_RTE-RRTMGP-181204_rad-irf_r1i1p1f1_gn.nc"), work)
        r = np.asarray(ref[rf][0], dtype=np.float64)                        # (expt, site, level)
        tst = np.stack(arrs, axis=0)
        if r.shape != tst.shape:
            raise RuntimeError(f"{rf}: reference shape {r.shape} vs test shape {tst.shape}")
        results.append(("rfmip " + rf, float(np.abs(tst - r).max()), threshold))
    return results


def atmosphere_from_input(dims, v, gases, tsi_ref=None):
    """synthetic.Atmosphere (the launcher-level pipeline's input) from the variables of an rte_rrtmgp_input file; used by the
    tests to push the acceptance inputs through another path."""
    from rte_rrtmgp_cpp_amd.synthetic import Atmosphere
    ncol, nlay = dims["x"]*dims["y"], dims["lay"]
    f2 = lambda n: np.ascontiguousarray(np.asarray(v[n][0], dtype=np.float64).reshape(-1, ncol))
    def vmr(g):
        a = np.asarray(v["vmr_" + g][0], dtype=np.float64)
        return np.ascontiguousarray(np.broadcast_to(a.reshape(-1, ncol) if a.ndim == 3 else (a[:, None] if a.ndim == 1 else a), (nlay, ncol)))
    p_lay = f2("p_lay")
    if "tsi" in v:
        tsi_scaling = np.asarray(v["tsi"][0], dtype=np.float64).reshape(ncol) / tsi_ref
    else:
        tsi_scaling = np.full(ncol, float(v["tsi_scaling"][0]) if "tsi_scaling" in v else 1.0)
    atm = dict(ncol=ncol, nlay=nlay, top_at_1=bool(p_lay[0, 0] < p_lay[-1, 0]), p_lay=p_lay, p_lev=f2("p_lev"), t_lay=f2("t_lay"),
               t_lev=f2("t_lev"), t_sfc=np.asarray(v["t_sfc"][0], dtype=np.float64).reshape(ncol).copy(),
               vmr={g: vmr(g) for g in gases if "vmr_" + g in v},
               emis_sfc=np.asarray(v["emis_sfc"][0], dtype=np.float64).reshape(ncol, -1).copy(),
               sfc_alb_dir=np.asarray(v["sfc_alb_dir"][0], dtype=np.float64).reshape(ncol, -1).copy(),
               sfc_alb_dif=np.asarray(v["sfc_alb_dif"][0], dtype=np.float64).reshape(ncol, -1).copy(),
               mu0=np.asarray(v["mu0"][0], dtype=np.float64).reshape(ncol).copy(), tsi_scaling=tsi_scaling)
    if "lwp" in v:
        for k in ("lwp", "iwp", "rel", "dei"):
            atm[k] = f2(k)
    return Atmosphere(**atm)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data", default=os.environ.get("RRTMGP_DATA"), help="rrtmgp-data checkout (default: $RRTMGP_DATA)")
    ap.add_argument("--case", default="both", choices=["allsky", "rfmip", "both"])
    ap.add_argument("--work", default="acceptance_work")
    a = ap.parse_args()
    if not a.data or not os.path.isdir(a.data):
        print("rrtmgp-data not found: set RRTMGP_DATA or pass --data (the data set is not part of this repository)")
        return 2
    results = []
    if a.case in ("allsky", "both"):
        results += case_allsky(a.data, os.path.join(a.work, "allsky"))
    if a.case in ("rfmip", "both"):
        results += case_rfmip(a.data, os.path.join(a.work, "rfmip"))
    failed = False
    for name, diff, thr in results:
        ok = diff <= thr
        failed |= not ok
        print(f"{name:24s} max abs difference {diff:.3e} W m-2  (threshold {thr:g})  {'ok' if ok else 'FAILED'}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
