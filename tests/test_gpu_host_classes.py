"""GPU tests of the C++ host layer (include/*.h class API + Radiation_solver + the test_rte_rrtmgp_gpu driver flow) through
the C entry point rrx_host_main of librte_rrtmgp_hip.so, on files in the reference's layout (synthetic_files.py).
The same inputs go through (a) the C++ classes, (b) the Python launcher-level pipeline on the HIP kernels and (c) the CPU
oracle; (a) exercises Gas_optics_rrtmgp_gpu's constructor reductions (an absent gas and its minor contributors are dropped)."""
import ctypes
import os

import numpy as np
import pytest

import cases
from rte_rrtmgp_cpp_amd import synthetic, synthetic_files, rrxio, pipeline

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTLIB = os.path.join(ROOT, "rte-rrtmgp-cpp_amd", "lib", "librte_rrtmgp_hip.so")
KW = dict(ngpt=48, nbnd=3, npres=12, nflav=4, nminor_lower=7, nminor_upper=4)


def run_driver(workdir, *flags, env=None):
    lib = ctypes.CDLL(HOSTLIB)
    argv = [b"test_rte_rrtmgp_gpu"] + [f.encode() for f in flags]
    arr = (ctypes.c_char_p * len(argv))(*argv)
    old = os.getcwd()
    saved = {}
    for k, v in (env or {}).items():
        saved[k] = os.environ.get(k); os.environ[k] = v
    try:
        os.chdir(workdir)
        rc = lib.rrx_host_main(len(argv), arr)
    finally:
        os.chdir(old)
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    return rc


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("rrx_case"))
    kl, ks = synthetic.make_kdist("lw", **KW), synthetic.make_kdist("sw", **KW)
    atm = synthetic.make_atmosphere(45, 60, nbnd_lw=KW["nbnd"], nbnd_sw=KW["nbnd"], clouds=True, seed=5)
    ll, ls = synthetic.make_cloud_lut(KW["nbnd"], "lw"), synthetic.make_cloud_lut(KW["nbnd"], "sw")
    synthetic_files.write_case(d, atm, kl, ks, ll, ls)
    return dict(dir=d, kl=kl, ks=ks, atm=atm, ll=ll, ls=ls)


def reference_fluxes(be, case, clouds):
    kl, ks = be.upload_kdist(case["kl"]), be.upload_kdist(case["ks"])
    atm = pipeline.upload_atmosphere(be, case["atm"])
    lw = pipeline.solve_lw(be, kl, atm, cloud_lut=be.upload_lut(case["ll"]) if clouds else None, keep=True)
    sw = pipeline.solve_sw(be, ks, atm, cloud_lut=be.upload_lut(case["ls"]) if clouds else None, delta_cloud=True, keep=True)
    N = be.to_numpy
    return dict(lw_flux_up=N(lw["flux_up"]), lw_flux_dn=N(lw["flux_dn"]), lw_flux_net=N(lw["flux_net"]),
                sw_flux_up=N(sw["flux_up"]), sw_flux_dn=N(sw["flux_dn"]), sw_flux_dn_dir=N(sw["flux_dn_dir"]), sw_flux_net=N(sw["flux_net"]),
                lw_tau=N(lw["tau"]), sw_tau=N(sw["tau"]), ssa=N(sw["ssa"]), g=N(sw["g"]),
                lay_source=N(lw["lay_src"]), lev_source=N(lw["lev_src"]),
                lw_gpt_up=N(lw["gpt_flux_up"]), sw_gpt_dn=N(sw["gpt_flux_dn"]))


def read_output(d):
    dims, v = rrxio.read(os.path.join(d, "rte_rrtmgp_output.nc"))
    return dims, {k: a[0].squeeze(axis=-2) if a[0].ndim >= 3 else a[0] for k, a in v.items()}


@pytest.mark.parametrize("clouds", [False, True])
@pytest.mark.parametrize("broadband", [True, False], ids=["broadband-solvers", "per-gpoint"])
def test_driver_matches_pipeline_and_oracle(case, clouds, broadband, hip_f64, oracle_f64):
    flags = (["--cloud-optics"] if clouds else []) + ([] if broadband else ["--no-broadband-solvers"])
    assert run_driver(case["dir"], *flags, "--output-optical") == 0
    _, out = read_output(case["dir"])
    hip = reference_fluxes(hip_f64, case, clouds)
    orc = reference_fluxes(oracle_f64, case, clouds)
    for k in ("lw_flux_up", "lw_flux_dn", "lw_flux_net", "sw_flux_up", "sw_flux_dn", "sw_flux_dn_dir", "sw_flux_net",
              "lw_tau", "sw_tau", "ssa", "g", "lay_source", "lev_source"):
        assert cases.rel_err(out[k], hip[k]) <= 1e-11, f"C++ classes vs launcher pipeline: {k}"
        assert cases.rel_err(out[k], orc[k]) <= (1e-7 if k.startswith("sw_flux") else 1e-9), f"C++ classes vs CPU oracle: {k}"


def test_driver_column_blocks_bands_and_broadband_mode(case, hip_f64):
    hip = reference_fluxes(hip_f64, case, True)
    # 45 columns in blocks of 7 (6 full blocks + a residual of 3), with band fluxes
    assert run_driver(case["dir"], "--cloud-optics", "--output-bnd-fluxes", env={"RRX_COL_BLOCK": "7"}) == 0
    _, out = read_output(case["dir"])
    for k in ("lw_flux_up", "lw_flux_dn", "sw_flux_up", "sw_flux_dn", "sw_flux_dn_dir"):
        assert cases.rel_err(out[k], hip[k]) <= 1e-11, k
    lims = case["kl"].band_lims_gpt
    for ib in range(KW["nbnd"]):
        want = hip["lw_gpt_up"][lims[ib, 0]-1:lims[ib, 1]].sum(axis=0)
        assert cases.rel_err(out["lw_bnd_flux_up"][ib], want) <= 1e-12
        want = hip["sw_gpt_dn"][lims[ib, 0]-1:lims[ib, 1]].sum(axis=0)
        assert cases.rel_err(out["sw_bnd_flux_dn"][ib], want) <= 1e-12
    assert cases.rel_err(out["lw_bnd_flux_up"].sum(axis=0), hip["lw_flux_up"]) <= 1e-12
    # broadband solvers (the CPU path's convention): same fluxes without per-g-point arrays
    assert run_driver(case["dir"], "--cloud-optics", "--broadband-solvers", "--no-delta-cloud") == 0
    assert run_driver(case["dir"], "--cloud-optics", "--broadband-solvers") == 0
    _, out = read_output(case["dir"])
    for k in ("lw_flux_up", "lw_flux_dn", "lw_flux_net", "sw_flux_up", "sw_flux_dn", "sw_flux_dn_dir", "sw_flux_net"):
        assert cases.rel_err(out[k], hip[k]) <= 1e-11, k
    # ... and with the fused one-kernel form of the broadband solvers forced (it needs >= 512 column groups by default)
    hip_f64.set_broadband_min_groups(1)
    try:
        assert run_driver(case["dir"], "--cloud-optics", env={"RRX_COL_BLOCK": "16"}) == 0
    finally:
        hip_f64.set_broadband_min_groups(512)
    _, out = read_output(case["dir"])
    for k in ("lw_flux_up", "lw_flux_dn", "lw_flux_net", "sw_flux_up", "sw_flux_dn", "sw_flux_dn_dir", "sw_flux_net"):
        assert cases.rel_err(out[k], hip[k]) <= 1e-11, k


def test_driver_error_behaviour(case):
    # same contract as the reference's main(): any exception -> message + exit status 1
    assert run_driver(case["dir"], "--bogus-option") == 1
    assert run_driver(case["dir"], "--aerosol-optics") == 1     # no aerosol_optics.nc / no rh, aermr* in this case's input
    assert run_driver(os.path.dirname(case["dir"])) == 1        # no input file there


def test_driver_ngpus_option(case):
    """--ngpus: one rank = the plain run (same bits); a malformed count is an error. More than one rank needs one GPU per rank
    (RCCL refuses two ranks on one device), so the sharded path is covered piecewise: rrx_column_range against
    sharding.column_range (tests/test_cabi.py), the pad / all-gather / place layout for world sizes 1..8 and a real one-rank
    communicator (tests/test_gpu_parity.py::test_rccl_allgather_fluxes_c_abi)."""
    assert run_driver(case["dir"], "--cloud-optics") == 0
    _, ref = read_output(case["dir"])
    assert run_driver(case["dir"], "--cloud-optics", "--ngpus=1") == 0
    _, one = read_output(case["dir"])
    assert run_driver(case["dir"], "--ngpus", "1", "--cloud-optics") == 0
    for k in ref:
        assert np.array_equal(ref[k], one[k]), k
    assert run_driver(case["dir"], "--ngpus=0") == 1


def test_delta_scale_on_lazy_zero_g_is_the_identity():
    """Optical_props_2str_gpu::delta_scale() on gas-only optical properties in the lazy g == 0 state must not
    hand the stale g array to the kernel (C++ self-check exported by the host library)."""
    lib = ctypes.CDLL(HOSTLIB)
    assert lib.rrx_host_selftest_delta_scale_gzero() == 0


def test_driver_heating_rates_and_async_mode(case, hip_f64):
    """--async (vertical ordering stated once, no per-solve read-backs) gives the same fluxes; --heating-rates
    writes -(g/cp) dF_net/dp per layer (checked against the formula on the driver's own fluxes)."""
    assert run_driver(case["dir"], "--cloud-optics") == 0
    _, ref = read_output(case["dir"])
    assert run_driver(case["dir"], "--cloud-optics", "--async", "--heating-rates") == 0
    _, out = read_output(case["dir"])
    for k in ("lw_flux_up", "lw_flux_dn", "lw_flux_net", "sw_flux_up", "sw_flux_dn", "sw_flux_dn_dir", "sw_flux_net"):
        assert np.array_equal(out[k], ref[k]), k
    p_lev = out["p_lev"]
    for kind in ("lw", "sw"):
        net = out[kind + "_flux_net"]
        want = -(9.80665/1004.64) * (net[1:] - net[:-1]) / (p_lev[1:] - p_lev[:-1])
        assert cases.rel_err(out[kind + "_heating_rate"], want) <= 1e-12, kind
    # shortwave only ever heats: the net downward flux cannot grow on the way down
    assert np.all(out["sw_heating_rate"] >= -1e-12 * np.abs(out["sw_heating_rate"]).max())


@pytest.fixture(scope="module")
def aerosol_case(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("rrx_aerosol_case"))
    kl, ks = synthetic.make_kdist("lw", **KW), synthetic.make_kdist("sw", **KW)
    atm = synthetic.make_atmosphere(45, 60, nbnd_lw=KW["nbnd"], nbnd_sw=KW["nbnd"], clouds=True, aerosols=True, seed=8)
    ll, ls = synthetic.make_cloud_lut(KW["nbnd"], "lw"), synthetic.make_cloud_lut(KW["nbnd"], "sw")
    la = synthetic.make_aerosol_lut(KW["nbnd"])
    synthetic_files.write_case(d, atm, kl, ks, ll, ls, la)
    return dict(dir=d, kl=kl, ks=ks, atm=atm, ll=ll, ls=ls, la=la)


@pytest.mark.parametrize("delta", [False, True])
def test_driver_aerosol_optics_matches_pipeline_and_oracle(aerosol_case, delta, hip_f64, oracle_f64):
    """--aerosol-optics [--delta-aerosol] through Aerosol_optics_gpu / Radiation_solver_shortwave, in column
    blocks of 7 (per-column mixing ratios are subset per block, profiles shared), against the launcher pipeline and the oracle."""
    c = aerosol_case
    flags = ["--aerosol-optics", "--cloud-optics", "--output-optical"] + (["--delta-aerosol"] if delta else [])
    assert run_driver(c["dir"], *flags, env={"RRX_COL_BLOCK": "7"}) == 0
    _, out = read_output(c["dir"])
    res = {}
    for be in (hip_f64, oracle_f64):
        r = pipeline.solve_sw(be, be.upload_kdist(c["ks"]), pipeline.upload_atmosphere(be, c["atm"]), cloud_lut=be.upload_lut(c["ls"]),
                              delta_cloud=True, aerosol_lut=be.upload_lut(c["la"]), delta_aerosol=delta, keep=True)
        res[be] = dict(sw_flux_up=r["flux_up"], sw_flux_dn=r["flux_dn"], sw_flux_dn_dir=r["flux_dn_dir"], sw_flux_net=r["flux_net"],
                       sw_tau=r["tau"], ssa=r["ssa"], g=r["g"])
        res[be] = {k: be.to_numpy(v) for k, v in res[be].items()}
    for k in res[hip_f64]:
        assert cases.rel_err(out[k], res[hip_f64][k]) <= 1e-11, f"C++ classes vs launcher pipeline: {k}"
        assert cases.rel_err(out[k], res[oracle_f64][k]) <= (1e-7 if k.startswith("sw_flux") else 1e-9), f"C++ classes vs CPU oracle: {k}"
    # single block = same numbers
    assert run_driver(c["dir"], *flags) == 0
    _, one = read_output(c["dir"])
    for k in res[hip_f64]:
        assert cases.rel_err(one[k], out[k]) <= 1e-12, k


def test_driver_on_netcdf4_files(aerosol_case, tmp_path):
    """the same case with every input file in NetCDF-4 (HDF5) form and NetCDF-4 output, converted back for the
    comparison: identical to the run on RRXB containers bit for bit."""
    c = aerosol_case
    flags = ["--aerosol-optics", "--cloud-optics", "--output-bnd-fluxes"]
    assert run_driver(c["dir"], *flags) == 0
    _, ref = read_output(c["dir"])
    lib = ctypes.CDLL(HOSTLIB)
    d = str(tmp_path)
    for f in os.listdir(c["dir"]):
        if f != "rte_rrtmgp_output.nc":
            assert lib.rrx_host_netcdf_convert(os.path.join(c["dir"], f).encode(), os.path.join(d, f).encode(), b"netcdf4") == 0
            with open(os.path.join(d, f), "rb") as fh:
                assert fh.read(4) == b"\x89HDF"
    assert run_driver(d, *flags, env={"RRX_OUTPUT_FORMAT": "netcdf4"}) == 0
    with open(os.path.join(d, "rte_rrtmgp_output.nc"), "rb") as fh:
        assert fh.read(4) == b"\x89HDF"
    assert lib.rrx_host_netcdf_convert(os.path.join(d, "rte_rrtmgp_output.nc").encode(), os.path.join(d, "out.rrxb").encode(), b"rrxb") == 0
    dims, v = rrxio.read(os.path.join(d, "out.rrxb"))
    out = {k: a[0].squeeze(axis=-2) if a[0].ndim >= 3 else a[0] for k, a in v.items()}
    assert set(out) == set(ref)
    for k in ref:
        assert np.array_equal(out[k], ref[k]), k


def test_acceptance_script_on_a_stand_in_data_tree(tmp_path, oracle_f64):
    """tools/acceptance.py (the reference's all-sky and RFMIP acceptance runs around this build's driver) end to
    end on a stand-in rrtmgp-data tree: NetCDF-4 coefficient files under the real names, an RFMIP-style input file with `units`
    attributes, and "reference" fluxes computed by the CPU oracle on the same problems. Thresholds are the reference's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("acceptance", os.path.join(ROOT, "tools", "acceptance.py"))
    acc = importlib.util.module_from_spec(spec); spec.loader.exec_module(acc)
    lib = ctypes.CDLL(HOSTLIB)
    data, tmp = str(tmp_path / "rrtmgp-data"), str(tmp_path / "tmp")
    os.makedirs(tmp)
    for sub in ("", acc.ALLSKY_REF, acc.RFMIP_REF, os.path.dirname(acc.RFMIP_IN)):
        os.makedirs(os.path.join(data, sub), exist_ok=True)

    def to_nc4(write, *args, name):
        write(os.path.join(tmp, "x.rrxb"), *args)
        assert lib.rrx_host_netcdf_convert(os.path.join(tmp, "x.rrxb").encode(), os.path.join(data, name).encode(), b"netcdf4") == 0

    kl, ks = synthetic.make_kdist("lw", **KW), synthetic.make_kdist("sw", **KW)
    ll, ls = synthetic.make_cloud_lut(KW["nbnd"], "lw"), synthetic.make_cloud_lut(KW["nbnd"], "sw")
    to_nc4(synthetic_files.write_kdist, kl, name=acc.GAS_LW); to_nc4(synthetic_files.write_kdist, ks, name=acc.GAS_SW)
    to_nc4(synthetic_files.write_cloud_lut, ll, name=acc.CLD_LW); to_nc4(synthetic_files.write_cloud_lut, ls, name=acc.CLD_SW)
    be = oracle_f64
    gases = list(kl.gas_names)

    def oracle_fluxes(dims, v, clouds):
        atm = pipeline.upload_atmosphere(be, acc.atmosphere_from_input(dims, v, gases, tsi_ref=float(ks.solar_source.sum())))
        lw = pipeline.solve_lw(be, be.upload_kdist(kl), atm, cloud_lut=be.upload_lut(ll) if clouds else None)
        sw = pipeline.solve_sw(be, be.upload_kdist(ks), atm, cloud_lut=be.upload_lut(ls) if clouds else None, delta_cloud=True)
        return {**{"lw_" + k: be.to_numpy(a) for k, a in lw.items()}, **{"sw_" + k: be.to_numpy(a) for k, a in sw.items()}}

    # all-sky reference files (lev, col), written as NetCDF-4
    dims, v = acc.allsky_input(KW["nbnd"], KW["nbnd"])
    f = oracle_fluxes(dims, v, True)
    d2 = dict(lev=dims["lev"], col=dims["x"])
    to_nc4(rrxio.write, d2, {"lw_flux_up": (f["lw_flux_up"], ["lev", "col"]), "lw_flux_dn": (f["lw_flux_dn"], ["lev", "col"])},
           name=os.path.join(acc.ALLSKY_REF, "rrtmgp-allsky-lw-no-aerosols.nc"))
    to_nc4(rrxio.write, d2, {"sw_flux_up": (f["sw_flux_up"], ["lev", "col"]), "sw_flux_dn": (f["sw_flux_dn"], ["lev", "col"]),
                             "sw_flux_dir": (f["sw_flux_dn_dir"], ["lev", "col"])},
           name=os.path.join(acc.ALLSKY_REF, "rrtmgp-allsky-sw-no-aerosols.nc"))

    # RFMIP-style input: 2 experiments x 5 sites x 20 layers, top first, concentrations scaled by their `units`
    n_expt, n_site, n_lay = 2, 5, 20
    base = synthetic.make_atmosphere(n_site, n_lay, nbnd_lw=KW["nbnd"], nbnd_sw=KW["nbnd"], top_at_1=True, seed=21)
    rd = dict(expt=n_expt, site=n_site, layer=n_lay, level=n_lay+1)
    e = np.arange(n_expt)[:, None, None]
    rv = {"pres_layer": (base.p_lay.T.copy(), ["site", "layer"]), "pres_level": (base.p_lev.T.copy(), ["site", "level"]),
          "temp_layer": (base.t_lay.T[None] + 2.0*e, ["expt", "site", "layer"]), "temp_level": (base.t_lev.T[None] + 2.0*e, ["expt", "site", "level"]),
          "surface_temperature": (base.t_sfc[None] + 2.0*e[:, :, 0], ["expt", "site"]),
          "surface_emissivity": (np.linspace(0.95, 0.99, n_site), ["site"]), "surface_albedo": (np.linspace(0.05, 0.3, n_site), ["site"]),
          "solar_zenith_angle": (np.linspace(10., 85., n_site), ["site"]), "total_solar_irradiance": (np.linspace(1300., 1400., n_site), ["site"]),
          "water_vapor": (base.vmr["h2o"].T[None] * (1. + 0.1*e) * 1e3, ["expt", "site", "layer"]),
          "ozone": (base.vmr["o3"].T[None] * (1. + 0.0*e) * 1e6, ["expt", "site", "layer"])}
    units = {"water_vapor": "1e-3", "ozone": "1e-6"}
    for gas, name in acc.RFMIP_GASES:
        if gas == "co":
            continue        # the stand-in k-distribution file lists co as a contributor with junk coefficients (absent-gas test)
        ref_v = float(base.vmr[gas][0, 0]) if gas in base.vmr else 1e-9
        rv[name] = (ref_v * 1e6 * (1. + 0.5*np.arange(n_expt)), ["expt"]); units[name] = "1e-6"
    to_nc4(rrxio.write, rd, rv, name=acc.RFMIP_IN)
    for name, u in units.items():
        assert lib.rrx_host_netcdf_put_attr(os.path.join(data, acc.RFMIP_IN).encode(), name.encode(), b"units", u.encode()) == 0
    assert acc.nc_attr(os.path.join(data, acc.RFMIP_IN), "methane_GM", "units") == "1e-6"
    got = {k: [] for k in ("rld", "rlu", "rsd", "rsu")}
    for expt, dims, v in acc.rfmip_inputs(data, tmp, KW["nbnd"], KW["nbnd"]):
        assert abs(float(v["vmr_co2"][0]) - 348.e-6*(1. + 0.5*expt)) < 1e-12 and v["p_lay"][0].shape == (n_lay, 1, n_site)
        f = oracle_fluxes(dims, v, False)
        for rf, name in (("rld", "lw_flux_dn"), ("rlu", "lw_flux_up"), ("rsd", "sw_flux_dn"), ("rsu", "sw_flux_up")):
            got[rf].append(f[name].T)
    for rf, arrs in got.items():
        to_nc4(rrxio.write, rd, {rf: (np.stack(arrs), ["expt", "site", "level"])},
               name=os.path.join(acc.RFMIP_REF, rf + "_Efx_RTE-RRTMGP-181204_rad-irf_r1i1p1f1_gn.nc"))

    results = acc.case_allsky(data, str(tmp_path / "work_allsky")) + acc.case_rfmip(data, str(tmp_path / "work_rfmip"))
    assert len(results) == 9
    for name, diff, thr in results:
        assert diff <= min(thr, 1e-6), (name, diff)


def test_driver_solves_differing_columns_in_order_of_surface_pressure(tmp_path, capfd):
    """--sort-columns (on by default): where neighbouring columns differ by more than a cell of the pressure grid the driver solves
    them in ascending order of surface pressure -- every per-column input permuted (fields, gases, clouds, aerosols, surface
    properties), every output written in the order of the input file. Same numbers as --no-sort-columns."""
    kl, ks = synthetic.make_kdist("lw", **KW), synthetic.make_kdist("sw", **KW)
    atm = synthetic.make_atmosphere(300, 60, nbnd_lw=KW["nbnd"], nbnd_sw=KW["nbnd"], clouds=True, aerosols=True, seed=11)
    rng = np.random.default_rng(3)
    f = rng.uniform(0.65, 1.35, atm.ncol)
    atm.p_lay = atm.p_lay * f; atm.p_lev = atm.p_lev * f
    dt = rng.uniform(-10., 10., atm.ncol)
    atm.t_lay = atm.t_lay + dt; atm.t_lev = atm.t_lev + dt; atm.t_sfc = atm.t_sfc + dt
    d = str(tmp_path)
    synthetic_files.write_case(d, atm, kl, ks, synthetic.make_cloud_lut(KW["nbnd"], "lw"), synthetic.make_cloud_lut(KW["nbnd"], "sw"),
                               synthetic.make_aerosol_lut(KW["nbnd"]))
    flags = ("--cloud-optics", "--aerosol-optics", "--output-bnd-fluxes", "--heating-rates")
    assert run_driver(d, *flags, "--no-sort-columns") == 0
    _, ref = read_output(d)
    assert "order of surface pressure" not in capfd.readouterr().out
    assert run_driver(d, *flags) == 0
    _, out = read_output(d)
    assert "order of surface pressure" in capfd.readouterr().out
    assert np.array_equal(out["p_lev"], ref["p_lev"])                       # the echo of the inputs keeps the file's order
    # (a column meets other neighbours in the sorted run, so it may take the windowed kernel in one run and the gather kernel in the
    #  other: 1e-15 apart in the optical depths, which the two-stream solver amplifies -- cases.Checker.sw_tol)
    n = 0
    for k in ref:
        if k.endswith(("_flux_up", "_flux_dn", "_flux_net", "_flux_dn_dir", "_heating_rate")):
            assert out[k].shape == ref[k].shape, k
            # (heating rates are differences of neighbouring net fluxes: 1e-13 of a flux is 1e-9 of its divergence)
            tol = 1e-6 if k.endswith("_heating_rate") else (1e-7 if k.startswith("sw_") else 1e-11)
            assert cases.rel_err(out[k], ref[k]) <= tol, k
            n += 1
    assert n >= 16
    # the same inside the solvers (Radiation_solver_*::set_column_sorting(1): device-side radix sort by surface pressure, inputs gathered,
    # 300 columns padded to 304, fluxes scattered back) instead of on the host before the upload
    assert run_driver(d, *flags, "--device-sort-columns") == 0
    _, dev = read_output(d)
    assert "order of surface pressure" not in capfd.readouterr().out
    for k in ref:
        if k.endswith(("_flux_up", "_flux_dn", "_flux_net", "_flux_dn_dir", "_heating_rate")):
            tol = 1e-6 if k.endswith("_heating_rate") else (1e-7 if k.startswith("sw_") else 1e-11)
            assert dev[k].shape == ref[k].shape and cases.rel_err(dev[k], ref[k]) <= tol, k
    # alike columns: nothing to sort
    atm2 = synthetic.make_atmosphere(300, 60, nbnd_lw=KW["nbnd"], nbnd_sw=KW["nbnd"], seed=11)
    synthetic_files.write_input(os.path.join(d, "rte_rrtmgp_input.nc"), atm2, KW["nbnd"], KW["nbnd"])
    assert run_driver(d) == 0
    assert "order of surface pressure" not in capfd.readouterr().out


def test_driver_on_tall_columns_and_narrow_bands(tmp_path, hip_f64):
    """The C++ driver on a shape beside the usual ones: 200 layers (the eight-wave / four-wave forms of the fused solvers) and
    8 g-points per band (band-aligned chunks of the windowed gas optics), against the Python pipeline on the same kernels."""
    kw = dict(ngpt=48, nbnd=6, npres=12, nflav=4, nminor_lower=7, nminor_upper=4)
    kl, ks = synthetic.make_kdist("lw", **kw), synthetic.make_kdist("sw", **kw)
    atm = synthetic.make_atmosphere(70, 200, nbnd_lw=kw["nbnd"], nbnd_sw=kw["nbnd"], seed=21)
    d = str(tmp_path)
    synthetic_files.write_case(d, atm, kl, ks)
    c = dict(dir=d, kl=kl, ks=ks, atm=atm, ll=None, ls=None)
    assert run_driver(d) == 0
    _, out = read_output(d)
    ref = reference_fluxes(hip_f64, c, clouds=False)
    for k in ("lw_flux_up", "lw_flux_dn", "lw_flux_net", "sw_flux_up", "sw_flux_dn", "sw_flux_dn_dir", "sw_flux_net"):
        assert cases.rel_err(out[k], ref[k]) <= (1e-7 if k.startswith("sw_") else 1e-11), k
