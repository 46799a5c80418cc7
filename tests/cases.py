"""Shared parity machinery: replay a golden fixture (tests/golden/*.npz, produced by oracle/make_golden.py from the
reference's own kernel text) through any backend and compare launcher by launcher."""
import glob
import hashlib
import os

import numpy as np

from rte_rrtmgp_cpp_amd import synthetic, pipeline

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MINI = dict(ngpt=32, nbnd=2, npres=10, nflav=3, nminor_lower=5, nminor_upper=3)


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def kdist_digest(kd):
    h = hashlib.sha256()
    for k in sorted(kd.__dict__):
        v = kd.__dict__[k]
        if isinstance(v, np.ndarray):
            h.update(k.encode()); h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


def arrays_digest(d):
    h = hashlib.sha256()
    for k in sorted(d):
        h.update(k.encode()); h.update(np.ascontiguousarray(d[k]).tobytes())
    return h.hexdigest()


def tall_inputs(seed, ncol, nlay, ngpt):
    """Seeded fp64 inputs of the tall solver fixtures (oracle/make_golden.py:tall_solver_case): the fixture stores their
    digest and the reference outputs. PCG64 + uniform() are stable across numpy versions."""
    rng = np.random.default_rng(seed)
    shp = (ngpt, nlay, ncol)
    d = dict(tau=10.0**rng.uniform(-6, 2, shp), lay=rng.uniform(5., 40., shp), lev=rng.uniform(5., 40., (ngpt, nlay+1, ncol)),
             emis=rng.uniform(0.8, 1.0, (ngpt, ncol)), ssrc=rng.uniform(5., 40., (ngpt, ncol)),
             ssa=rng.uniform(0., 1., shp), g=rng.uniform(-0.3, 0.9, shp), mu0=rng.uniform(0.05, 1.0, ncol),
             adif=rng.uniform(0., 0.6, (ngpt, ncol)), inc=rng.uniform(0., 5., (ngpt, ncol)))
    d["adir"] = np.ascontiguousarray(np.repeat(rng.uniform(0., 0.6, ncol)[None, :], ngpt, axis=0))
    d["tau"][0, 0, :] = 0.0
    d["ssa"][1, ::7, :] = 1.0
    d["ssa"][2, ::5, :] = 0.0
    return d


def rel_err(a, b, floor=1e-6):
    """max |a-b| / (|b| + floor*max|b|): relative error that does not blow up at the zeros of b.
    fp32 comparisons use floor = 1e-2: single-precision cancellation in the source terms leaves an ABSOLUTE error of
    a few eps x (largest flux), so values many orders below the largest one carry no significant digits in either
    the reference or the HIP result."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return 0.0
    scale = np.abs(b) + floor*np.max(np.abs(b)) + 1e-300
    err = float(np.max(np.abs(a - b) / scale))
    _note_worst(err)
    return err


# worst relative error seen by each test (tests/conftest.py prints the table at the end of the run, so that a regression
# inside a tolerance is visible in the log)
WORST = {}


def _note_worst(err):
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    if test:
        WORST[test] = max(WORST.get(test, 0.0), err)


class Checker:
    def __init__(self, be, tol):
        self.be, self.tol, self.worst = be, tol, {}
        self.floor = 1e-6 if be.np_dtype == np.float64 else 1e-2
        # fp64 SW two-stream outputs of the HIP path: FMA contraction and Newton reciprocals re-round the near-resonant
        # (k*mu0 ~ 1) and conservative-scattering (k_min clamp) cells, which are ill-conditioned in the reference too.
        self.sw_tol = max(tol, 1e-7) if (be.np_dtype == np.float64 and be.name.startswith("hip")) else tol

    def close(self, name, got, want, tol=None):
        e = rel_err(self.be.to_numpy(got), want, self.floor)
        self.worst[name] = e
        if tol is None and name.startswith(("sw_flux", "sw_dif_flux", "sw_gpt_flux", "sw_bb")):
            tol = self.sw_tol
        assert e <= (tol or self.tol), f"{name}: rel err {e:.3e} > {(tol or self.tol):.1e} ({self.be.name})"

    def exact(self, name, got, want):
        g = self.be.to_numpy(got)
        assert np.array_equal(g.astype(np.int64), np.asarray(want).astype(np.int64)), f"{name}: integer mismatch ({self.be.name})"

    def same_bits(self, name, got, want):
        g = self.be.to_numpy(got); w = self.be.to_numpy(want)
        assert g.dtype == w.dtype and np.array_equal(g.view(np.uint8), w.view(np.uint8)), f"{name}: not bit-identical ({self.be.name})"


def dtype_of(G):
    return G["lw_tau"].dtype


def run_chain_case(be, G, tol):
    """gas optics -> Planck -> LW solver -> broadband; gas optics + Rayleigh -> combine -> SW solver -> broadband."""
    ck = Checker(be, tol)
    ncol, nlay, top_at_1, _ = [int(x) for x in G["meta"]]
    up = be.asarray
    play, tlay, tlev, tsfc = up(G["p_lay"]), up(G["t_lay"]), up(G["t_lev"]), up(G["t_sfc"])
    col_dry, col_gas = up(G["col_dry"]), up(G["col_gas"])

    # the small kernels the reference keeps in its host classes: checked against the fixture's numpy inputs
    if hasattr(be, "get_col_dry"):
        ck.close("col_dry", be.get_col_dry(up(G["vmr_h2o"]), up(G["p_lev"])), G["col_dry"], tol=max(tol, 1e-12))

    for kind in ("lw", "sw"):
        kd0 = synthetic.make_kdist(kind, **MINI)
        assert kdist_digest(kd0) == str(G[f"{kind}_kdist_digest"]), "synthetic k-distribution drifted: regenerate goldens"
        kd = be.upload_kdist(kd0)
        vmr = {n: up(G[f"vmr_{n}"]) for n in kd0.gas_names}
        ck.close(f"{kind}_col_gas", be.fill_gases(kd, vmr, col_dry), G["col_gas"], tol=max(tol, 1e-12))
        it = be.interpolation(kd, play, tlay, col_gas)
        for k in ("jtemp", "jpress", "tropo", "jeta"):
            ck.exact(f"{kind}_it_{k}", it[k], G[f"{kind}_it_{k}"])
        for k in ("col_mix", "fminor", "fmajor"):
            ck.close(f"{kind}_it_{k}", it[k], G[f"{kind}_it_{k}"])
        # downstream kernels are fed the REFERENCE interpolation state so that errors do not compound
        it = {k: up(G[f"{kind}_it_{k}"]) for k in it}
        if kind == "lw":
            tau = be.zeros((kd.ngpt, nlay, ncol))
            be.compute_tau_absorption(kd, it, play, tlay, col_gas, tau)
            ck.close("lw_tau", tau, G["lw_tau"])
            if hasattr(be, "compute_tau_absorption_set"):       # store form == add form on a zeroed tau, bit for bit
                t2 = be.compute_tau_absorption_set(kd, it, play, tlay, col_gas, be.empty(tau.shape))
                ck.same_bits("lw_tau_set", t2, tau)
            src = be.compute_planck_source(kd, it, tlay, tlev, tsfc, nlay if top_at_1 else 1)
            for k in ("lay_src", "lev_src", "sfc_src", "sfc_src_jac"):
                ck.close("lw_" + k, src[k], G["lw_" + k])
            sec = be.lw_secants_array(ncol, kd.ngpt, 1, 4, up(pipeline.GAUSS_DS))
            ck.close("lw_secants", sec, G["lw_secants"])
            fl = be.lw_solver_noscat(bool(top_at_1), up(G["lw_secants"]), up(np.array([1.0])), up(G["lw_tau"]),
                                     up(G["lw_lay_src"]), up(G["lw_lev_src"]), up(G["lw_sfc_emis_gpt"]), up(G["lw_sfc_src"]))
            ck.close("lw_gpt_flux_up", fl["flux_up"], G["lw_gpt_flux_up"])
            ck.close("lw_gpt_flux_dn", fl["flux_dn"], G["lw_gpt_flux_dn"])
            fu = be.sum_broadband(up(G["lw_gpt_flux_up"])); fd = be.sum_broadband(up(G["lw_gpt_flux_dn"]))
            ck.close("lw_flux_up", fu, G["lw_flux_up"]); ck.close("lw_flux_dn", fd, G["lw_flux_dn"])
            ck.close("lw_flux_net", be.net_broadband_precalc(up(G["lw_flux_dn"]), up(G["lw_flux_up"])), G["lw_flux_net"])
            # broadband mode of the solver == sum of the per-g-point fluxes (CPU/Fortran semantics)
            if be.name != "ref":
                bb = be.lw_solver_noscat(bool(top_at_1), up(G["lw_secants"]), up(np.array([1.0])), up(G["lw_tau"]),
                                         up(G["lw_lay_src"]), up(G["lw_lev_src"]), up(G["lw_sfc_emis_gpt"]), up(G["lw_sfc_src"]),
                                         do_broadband=True)
                ck.close("lw_bb_flux_up", bb["flux_up"], G["lw_flux_up"], tol=max(tol, 1e-12))
                ck.close("lw_bb_flux_dn", bb["flux_dn"], G["lw_flux_dn"], tol=max(tol, 1e-12))
        else:
            tau_abs = be.zeros((kd.ngpt, nlay, ncol))
            be.compute_tau_absorption(kd, it, play, tlay, col_gas, tau_abs)
            ck.close("sw_tau_abs", tau_abs, G["sw_tau_abs"])
            ck.close("sw_tau_ray", be.compute_tau_rayleigh(kd, it, col_dry, col_gas), G["sw_tau_ray"])
            tau, ssa, g = be.combine_abs_and_rayleigh(up(G["sw_tau_abs"]), up(G["sw_tau_ray"]))
            ck.close("sw_tau", tau, G["sw_tau"]); ck.close("sw_ssa", ssa, G["sw_ssa"]); ck.close("sw_g", g, G["sw_g"])
            if hasattr(be, "gas_optics_sw_fused"):
                t2 = be.empty(tau.shape); w2 = be.empty(tau.shape); g2 = be.empty(tau.shape)
                be.gas_optics_sw_fused(kd, it, play, tlay, col_gas, col_dry, t2, w2, g2)
                ck.close("sw_fused_tau", t2, G["sw_tau"]); ck.close("sw_fused_ssa", w2, G["sw_ssa"]); ck.close("sw_fused_g", g2, G["sw_g"])
            fl = be.sw_solver_2stream(bool(top_at_1), up(G["sw_tau"]), up(G["sw_ssa"]), up(G["sw_g"]), up(G["mu0"]),
                                      up(G["sw_alb_dir"]), up(G["sw_alb_dif"]), up(G["sw_toa_src"]))
            ck.close("sw_gpt_flux_up", fl["flux_up"], G["sw_gpt_flux_up"])
            ck.close("sw_gpt_flux_dn", fl["flux_dn"], G["sw_gpt_flux_dn"])
            ck.close("sw_gpt_flux_dir", fl["flux_dir"], G["sw_gpt_flux_dir"])
            ck.close("sw_flux_up", be.sum_broadband(up(G["sw_gpt_flux_up"])), G["sw_flux_up"])
            if be.name != "ref":
                bb = be.sw_solver_2stream(bool(top_at_1), up(G["sw_tau"]), up(G["sw_ssa"]), up(G["sw_g"]), up(G["mu0"]),
                                          up(G["sw_alb_dir"]), up(G["sw_alb_dif"]), up(G["sw_toa_src"]), do_broadband=True)
                ck.close("sw_bb_flux_up", bb["flux_up"], G["sw_flux_up"], tol=max(tol, 1e-12))
                ck.close("sw_bb_flux_dn", bb["flux_dn"], G["sw_flux_dn"], tol=max(tol, 1e-12))
                ck.close("sw_bb_flux_dir", bb["flux_dir"], G["sw_flux_dir"], tol=max(tol, 1e-12))
    return ck.worst


def run_random_case(be, G, tol):
    ck = Checker(be, tol)
    ncol, nlay, top_at_1, _ = [int(x) for x in G["meta"]]
    up = be.asarray
    ngpt = G["lw_tau"].shape[0]
    sec = be.lw_secants_array(ncol, ngpt, 1, 4, up(pipeline.GAUSS_DS))
    fl = be.lw_solver_noscat(bool(top_at_1), sec, up(np.array([1.0])), up(G["lw_tau"]), up(G["lw_lay_src"]), up(G["lw_lev_src"]),
                             up(G["lw_emis"]), up(G["lw_sfc_src"]), do_jacobians=True, sfc_src_jac=up(G["lw_sfc_src_jac"]))
    ck.close("lw_flux_up", fl["flux_up"], G["lw_flux_up"]); ck.close("lw_flux_dn", fl["flux_dn"], G["lw_flux_dn"])
    ck.close("lw_flux_up_jac", fl["flux_up_jac"], G["lw_flux_up_jac"])

    args = (bool(top_at_1), up(G["lw_tau"]), up(G["sw_ssa"]), up(G["sw_g"]), up(G["mu0"]), up(G["sw_alb_dir"]), up(G["sw_alb_dif"]), up(G["sw_inc_dir"]))
    fs = be.sw_solver_2stream(*args)
    ck.close("sw_flux_up", fs["flux_up"], G["sw_flux_up"]); ck.close("sw_flux_dn", fs["flux_dn"], G["sw_flux_dn"])
    ck.close("sw_flux_dir", fs["flux_dir"], G["sw_flux_dir"])
    fd = be.sw_solver_2stream(*args, inc_flux_dif=up(G["sw_inc_dif"]))
    ck.close("sw_dif_flux_up", fd["flux_up"], G["sw_dif_flux_up"]); ck.close("sw_dif_flux_dn", fd["flux_dn"], G["sw_dif_flux_dn"])

    t1, w1, g1 = up(G["lw_tau"].copy()), up(G["sw_ssa"].copy()), up(G["sw_g"].copy())
    be.increment_2stream_by_2stream(t1, w1, g1, up(G["op_t2"]), up(G["op_w2"]), up(G["op_g2"]))
    ck.close("op_inc2_tau", t1, G["op_inc2_tau"]); ck.close("op_inc2_ssa", w1, G["op_inc2_ssa"]); ck.close("op_inc2_g", g1, G["op_inc2_g"])
    t1 = up(G["lw_tau"].copy()); be.increment_1scalar_by_1scalar(t1, up(G["op_t2"])); ck.close("op_inc1_tau", t1, G["op_inc1_tau"])
    lims = up(G["op_lims"])
    t1, w1, g1 = up(G["lw_tau"].copy()), up(G["sw_ssa"].copy()), up(G["sw_g"].copy())
    be.inc_2stream_by_2stream_bybnd(t1, w1, g1, up(G["op_tb"]), up(G["op_wb"]), up(G["op_gb"]), lims)
    ck.close("op_incb2_tau", t1, G["op_incb2_tau"]); ck.close("op_incb2_ssa", w1, G["op_incb2_ssa"]); ck.close("op_incb2_g", g1, G["op_incb2_g"])
    t1 = up(G["lw_tau"].copy()); be.inc_1scalar_by_1scalar_bybnd(t1, up(G["op_tb"]), lims); ck.close("op_incb1_tau", t1, G["op_incb1_tau"])
    t1, w1, g1 = up(G["lw_tau"].copy()), up(G["sw_ssa"].copy()), up(G["sw_g"].copy())
    be.delta_scale_2str_k(t1, w1, g1)
    ck.close("op_ds_tau", t1, G["op_ds_tau"]); ck.close("op_ds_ssa", w1, G["op_ds_ssa"]); ck.close("op_ds_g", g1, G["op_ds_g"])
    return ck.worst


def run_tall_case(be, G, tol):
    """60- / 140-layer random-input solver runs against the reference kernel text: the per-g-point kernels and, on backends
    with a broadband mode, the fused forms against the in-order g-point sums of the same fixture."""
    ck = Checker(be, tol)
    ncol, nlay, top_at_1, seed, ngpt = [int(x) for x in G["meta"]]
    I = tall_inputs(seed, ncol, nlay, ngpt)
    assert arrays_digest(I) == str(G["inputs_digest"]), "seeded inputs drifted: regenerate goldens"
    up = be.asarray
    sec = be.lw_secants_array(ncol, ngpt, 1, 4, up(pipeline.GAUSS_DS)); w = up(np.array([1.0]))
    lw = (bool(top_at_1), sec, w, up(I["tau"]), up(I["lay"]), up(I["lev"]), up(I["emis"]), up(I["ssrc"]))
    fl = be.lw_solver_noscat(*lw)
    ck.close("lw_flux_up", fl["flux_up"], G["lw_flux_up"]); ck.close("lw_flux_dn", fl["flux_dn"], G["lw_flux_dn"])
    sw = (bool(top_at_1), up(I["tau"]), up(I["ssa"]), up(I["g"]), up(I["mu0"]), up(I["adir"]), up(I["adif"]), up(I["inc"]))
    fs = be.sw_solver_2stream(*sw)
    for k in ("up", "dn", "dir"):
        ck.close("sw_flux_" + k, fs["flux_" + k], G["sw_flux_" + k])
    if be.name != "ref":
        seqsum = lambda a: np.add.reduce(a, axis=0)          # ascending g-point order = sum_broadband's
        bb = be.lw_solver_noscat(*lw, do_broadband=True)
        ck.close("lw_bb_flux_up", bb["flux_up"], seqsum(G["lw_flux_up"]), tol=max(tol, 1e-12))
        ck.close("lw_bb_flux_dn", bb["flux_dn"], seqsum(G["lw_flux_dn"]), tol=max(tol, 1e-12))
        bb = be.sw_solver_2stream(*sw, do_broadband=True)
        for k in ("up", "dn", "dir"):
            ck.close("sw_bb_flux_" + k, bb["flux_" + k], seqsum(G["sw_flux_" + k]))
        # clear-sky form: g identically zero (HIP: no g array at all -> the hand-folded two-stream coefficients)
        g0 = None if getattr(be, "supports_null_g", False) else up(np.zeros_like(I["g"]))
        b0 = be.sw_solver_2stream(sw[0], sw[1], sw[2], g0, *sw[4:], do_broadband=True)
        for k in ("up", "dn", "dir"):
            ck.close("sw_bb_g0_flux_" + k, b0["flux_" + k], G["sw_g0_bb_" + k])
    return ck.worst


def run_chainbb_case(be, G, tol, **modes):
    """Atmosphere -> broadband fluxes through pipeline.solve_lw / solve_sw in the backend's default (product) chain, against
    the reference kernel text's chain at 140 layers."""
    ck = Checker(be, tol)
    ncol, nlay, top_at_1, _ = [int(x) for x in G["meta"]]
    vmr = {k[4:]: G[k] for k in G.files if k.startswith("vmr_")}
    atm0 = synthetic.make_atmosphere(ncol, nlay, nbnd_lw=2, nbnd_sw=2, top_at_1=bool(top_at_1), seed=7)
    for k in ("p_lay", "p_lev", "t_lay", "t_lev", "t_sfc", "mu0"):
        assert np.array_equal(getattr(atm0, k), G[k]), f"synthetic atmosphere drifted ({k}): regenerate goldens"
    for n in vmr:
        assert np.array_equal(atm0.vmr[n], vmr[n]), n
    atm = pipeline.upload_atmosphere(be, atm0)
    for kind in ("lw", "sw"):
        kd0 = synthetic.make_kdist(kind, **MINI)
        assert kdist_digest(kd0) == str(G[f"{kind}_kdist_digest"]), "synthetic k-distribution drifted: regenerate goldens"
        kd = be.upload_kdist(kd0)
        r = (pipeline.solve_lw if kind == "lw" else pipeline.solve_sw)(be, kd, atm, col_dry=be.asarray(G["col_dry"]), **modes)
        t = None if kind == "sw" else max(tol, 1e-12)          # SW: Checker.sw_tol
        ck.close(f"{kind}_flux_up", r["flux_up"], G[f"{kind}_flux_up"], tol=t)
        ck.close(f"{kind}_flux_dn", r["flux_dn"], G[f"{kind}_flux_dn"], tol=t)
        if kind == "sw":
            ck.close("sw_flux_dir", r["flux_dn_dir"], G["sw_flux_dir"])
        else:
            ck.close("lw_flux_net", r["flux_net"], G["lw_flux_net"], tol=max(tol, 1e-11))
    return ck.worst


def run_cloud_case(be, G, tol):
    """rrx_cloud_optics_{2str,1scl} / the restatement against the reference's own CPU class (src/Cloud_optics.cpp:29-232)."""
    ck = Checker(be, tol)
    lut = synthetic.make_cloud_lut(int(G["lut_nbnd"]), str(G["lut_kind"]))
    assert optics_lut_digest(lut) == str(G["lut_digest"]), "synthetic cloud LUT drifted: regenerate goldens"
    dt = G["clwp"].dtype
    l = be.upload_lut({k: (v.astype(dt) if isinstance(v, np.ndarray) else v) for k, v in lut.items()})
    up = be.asarray
    args = (l, up(G["clwp"]), up(G["ciwp"]), up(G["reliq"]), up(G["deice"]))
    t2, w2, g2 = be.cloud_optics_2str(*args)
    ck.close("cloud_tau_2str", t2, G["tau_2str"]); ck.close("cloud_ssa_2str", w2, G["ssa_2str"]); ck.close("cloud_g_2str", g2, G["g_2str"])
    ck.close("cloud_tau_1scl", be.cloud_optics_1scl(*args), G["tau_1scl"])
    return ck.worst


def optics_lut_digest(lut):
    h = hashlib.sha256()
    for k in sorted(lut):
        h.update(k.encode()); h.update(np.ascontiguousarray(np.asarray(lut[k], dtype=np.float64)).tobytes())
    return h.hexdigest()


def run_aerosol_case(be, G, tol, tmp_dir):
    """rrx_aerosol_optics / the restatement against the reference's own CPU class (src/Aerosol_optics.cpp:24-224) on the real
    CAMS tables and on synthetic ones."""
    ck = Checker(be, tol)
    lut = real_aerosol_lut(tmp_dir) if str(G["table"]) == "real" else synthetic.make_aerosol_lut(5)
    assert optics_lut_digest(lut) == str(G["lut_digest"]), "aerosol tables drifted: regenerate goldens"
    dt = G["rh"].dtype
    l = be.upload_lut({k: v.astype(dt) for k, v in lut.items()})
    up = be.asarray
    tau, ssa, g = be.aerosol_optics(l, [up(G["aermr%02d" % i]) for i in range(1, 12)], up(G["rh"]), up(G["p_lev"]))
    ck.close("aerosol_tau", tau, G["tau"]); ck.close("aerosol_ssa", ssa, G["ssa"]); ck.close("aerosol_g", g, G["g"])
    return ck.worst


def run_reference_classes_case(which, G, tol, sw_tol):
    """The reference's own CPU classes (unmodified src/{Rte_lw,Rte_sw,Fluxes,Optical_props,Source_functions}.cpp) on top of a
    library exporting the 19 bind(C) kernels -- `which` = "oracle" (the restatement) or "hip" (the product's CPU boundary) --
    replaying a tall fixture: (i) one g-point per band, so that the by-band surface arrays of the class API are the fixture's
    per-g-point ones: fluxes against the reference KERNEL TEXT; (ii) two bands, clouds added by band through add_to /
    delta_scale, broadband mode, two quadrature angles: against the oracle called directly."""
    import cpu_boundary
    import oracle_py
    ncol, nlay, top_at_1, seed, ngpt = [int(x) for x in G["meta"]]
    I = tall_inputs(seed, ncol, nlay, ngpt)
    worst = {}

    def close(name, got, want, t):
        e = rel_err(got, want)
        worst[name] = e
        assert e <= t, f"{name}: rel err {e:.3e} > {t:.1e} (reference classes on {which})"

    # (i) against the fixture
    lims1 = np.stack([np.arange(1, ngpt+1), np.arange(1, ngpt+1)], axis=1)
    o = cpu_boundary.run_reference_classes(which, "lw", (ncol, nlay, ngpt, ngpt), [I["tau"], I["lay"], I["lev"], I["ssrc"], I["emis"].T],
                                           lims1, top_at_1)
    close("cls_lw_gpt_up", o[1], G["lw_flux_up"], tol); close("cls_lw_gpt_dn", o[2], G["lw_flux_dn"], tol)
    close("cls_lw_bb_up", o[3], np.add.reduce(G["lw_flux_up"], axis=0), max(tol, 1e-12))
    close("cls_lw_bb_net", o[5], np.add.reduce(G["lw_flux_dn"], axis=0) - np.add.reduce(G["lw_flux_up"], axis=0), max(tol, 1e-11))
    o = cpu_boundary.run_reference_classes(which, "sw", (ncol, nlay, ngpt, ngpt),
                                           [I["tau"], I["ssa"], I["g"], I["mu0"], I["inc"], I["adir"].T, I["adif"].T], lims1, top_at_1)
    for i, k in enumerate(("up", "dn", "dir")):
        close("cls_sw_gpt_" + k, o[3+i], G["sw_flux_" + k], sw_tol)
    close("cls_sw_bb_dir", o[8], np.add.reduce(G["sw_flux_dir"], axis=0), sw_tol)

    # (ii) against the oracle called directly
    orc = oracle_py.CpuKernels("oracle", np.float64)
    rng = np.random.default_rng(seed + 1000)
    lims2 = np.array([[1, 2], [3, ngpt]], dtype=np.int32)
    cld_t = 10.0**rng.uniform(-3, 1, (2, nlay, ncol)); cld_w = rng.uniform(0, 1, cld_t.shape); cld_g = rng.uniform(0, .9, cld_t.shape)
    emis_b = rng.uniform(.8, 1., (ncol, 2)); inc = rng.uniform(0., 5., (ngpt, ncol))
    o = cpu_boundary.run_reference_classes(which, "lw", (ncol, nlay, ngpt, 2), [I["tau"], I["lay"], I["lev"], I["ssrc"], emis_b, inc, cld_t],
                                           lims2, top_at_1, broadband=True, n_angles=2, has_inc=True, has_bybnd=True)
    tau = I["tau"].copy(); orc.inc_1scalar_by_1scalar_bybnd(tau, cld_t, lims2)
    emis_g = orc.expand_and_transpose(lims2, np.ascontiguousarray(emis_b), ngpt)
    sec = orc.lw_secants_array(ncol, ngpt, 2, 4, pipeline.GAUSS_DS)
    want = orc.lw_solver_noscat(bool(top_at_1), sec, np.ascontiguousarray(pipeline.GAUSS_WTS[1, :2]), tau, I["lay"], I["lev"], emis_g, I["ssrc"],
                                inc_flux=inc, do_broadband=True)
    close("cls_lw_tau_incremented", o[0], tau, tol)
    close("cls_lw_2ang_bb_up", o[1], want["flux_up"], max(tol, 1e-12)); close("cls_lw_2ang_bb_dn", o[2], want["flux_dn"], max(tol, 1e-12))
    adir_b = rng.uniform(0., .6, (ncol, 2)); adif_b = rng.uniform(0., .6, (ncol, 2)); inc_dif = rng.uniform(0., 1., (ngpt, ncol))
    o = cpu_boundary.run_reference_classes(which, "sw", (ncol, nlay, ngpt, 2),
                                           [I["tau"], I["ssa"], I["g"], I["mu0"], I["inc"], adir_b, adif_b, inc_dif, cld_t, cld_w, cld_g],
                                           lims2, top_at_1, broadband=True, has_inc=True, has_bybnd=True, delta=True)
    t, w, g = I["tau"].copy(), I["ssa"].copy(), I["g"].copy()
    ct, cw, cg = cld_t.copy(), cld_w.copy(), cld_g.copy()
    orc.delta_scale_2str_k(ct, cw, cg); orc.inc_2stream_by_2stream_bybnd(t, w, g, ct, cw, cg, lims2)
    want = orc.sw_solver_2stream(bool(top_at_1), t, w, g, I["mu0"], orc.expand_and_transpose(lims2, np.ascontiguousarray(adir_b), ngpt),
                                 orc.expand_and_transpose(lims2, np.ascontiguousarray(adif_b), ngpt), I["inc"], inc_flux_dif=inc_dif, do_broadband=True)
    close("cls_sw_tau", o[0], t, max(tol, 1e-12)); close("cls_sw_ssa", o[1], w, max(tol, 1e-12)); close("cls_sw_g", o[2], g, max(tol, 1e-12))
    for i, k in enumerate(("up", "dn", "dir")):
        close("cls_sw_cld_bb_" + k, o[3+i], want["flux_" + k], sw_tol)
    return worst


def run_glue_case(be, G, tol):
    """Stand-alone apply_BC x3 and transposes (data movement: bit-exact) and the LW incident-flux convention."""
    ck = Checker(be, tol)
    ncol, nlay, top_at_1, _ = [int(x) for x in G["meta"]]
    up = be.asarray
    if be.name != "oracle":      # the CPU boundary has no stand-alone apply_BC / transposes (commented out of rrtmgp_kernels.h)
        ck.same_bits("bc_0", be.apply_BC(nlay, bool(top_at_1), up(G["bc_base"].copy())), up(G["bc_0"]))
        ck.same_bits("bc_gpt", be.apply_BC(nlay, bool(top_at_1), up(G["bc_base"].copy()), up(G["bc_inc"])), up(G["bc_gpt"]))
        ck.same_bits("bc_fac", be.apply_BC(nlay, bool(top_at_1), up(G["bc_base"].copy()), up(G["bc_inc"]), up(G["bc_factor"])), up(G["bc_fac"]))
        ck.same_bits("ro_321", be.reorder123x321(up(G["ro_a3"])), up(G["ro_321"]))
        ck.same_bits("ro_21", be.reorder12x21(up(G["ro_a2"])), up(G["ro_21"]))
    # the reference's CUDA text turns an incident flux F into a boundary flux F/2 (rte_solver_kernels.cu:160,
    # 189-190); this build (and the CPU restatement) keeps F, the CPU/Fortran semantics. So the fixture, computed by the
    # reference text with incident flux F, must be reproduced here with F/2 -- the factor of two is explicit and pinned.
    ngpt = G["lw_tau"].shape[0]
    sec = be.lw_secants_array(ncol, ngpt, 1, 4, up(pipeline.GAUSS_DS))
    half = (G["lw_inc"] * G["lw_inc"].dtype.type(0.5))
    fl = be.lw_solver_noscat(bool(top_at_1), sec, up(np.array([1.0])), up(G["lw_tau"]), up(G["lw_lay_src"]), up(G["lw_lev_src"]),
                             up(G["lw_emis"]), up(G["lw_sfc_src"]), inc_flux=up(half))
    ck.close("lw_inc_flux_dn", fl["flux_dn"], G["lw_inc_flux_dn"]); ck.close("lw_inc_flux_up", fl["flux_up"], G["lw_inc_flux_up"])
    top = 0 if top_at_1 else nlay
    got_top = be.to_numpy(fl["flux_dn"])[:, top, :]
    assert rel_err(got_top, half) <= 10*tol, "flux_dn at the top of the domain must equal the incident flux handed in"
    return ck.worst


def real_aerosol_lut(tmp_dir):
    """The tables of the reference tree's data/aerosol_optics.nc (data fixture tests/golden/aerosol_optics.nc), read through
    the NetCDF-4 backend of the host library (converted to RRXB, then rrxio)."""
    import ctypes
    from rte_rrtmgp_cpp_amd import rrxio, synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = ctypes.CDLL(os.path.join(root, "rte-rrtmgp-cpp_amd", "lib", "librte_rrtmgp_hip.so"))
    out = os.path.join(str(tmp_dir), "aerosol_optics.rrxb")
    assert lib.rrx_host_netcdf_convert(os.path.join(root, "tests", "golden", "aerosol_optics.nc").encode(), out.encode(), b"rrxb") == 0
    _, v = rrxio.read(out)
    return synthetic.aerosol_lut_from_file_vars(v)


def aerosol_optics_numpy(lut, aermr, rh, plev):
    """Independent, vectorised evaluation of /root/reference/src/Aerosol_optics.cpp:38-224 (not the oracle's loop nest): used to
    pin the oracle's restatement. Returns tau, ssa, g (nbnd, nlay, ncol)."""
    nlay, ncol = rh.shape
    eps = np.finfo(rh.dtype).eps
    dpg = np.abs(plev[:-1] - plev[1:]) / rh.dtype.type(9.81)
    ihum = np.minimum(np.searchsorted(lut["rh_upper"], rh, side="left"), len(lut["rh_upper"]) - 1)     # first bound >= rh
    species = [(1, "philic", 1), (2, "philic", 2), (3, "philic", 3), (4, "phobic", 1), (5, "phobic", 8), (6, "phobic", 6),
               (8, "phobic", 10), (7, "philic", 4), (9, "phobic", 11), (10, "phobic", 11), (11, "philic", 5)]
    nbnd = lut["mext_phobic"].shape[-1]
    tau = np.zeros((nbnd, nlay, ncol), rh.dtype); ts = np.zeros_like(tau); tsg = np.zeros_like(tau)
    for im, kind, col in species:
        m = aermr[im-1]
        m = np.broadcast_to(m if m.ndim == 2 else m[:, None], (nlay, ncol))
        if kind == "phobic":
            tab = lambda n: lut[n + "_phobic"][col-1][:, None, None]
        else:
            tab = lambda n: np.moveaxis(lut[n + "_philic"][col-1][ihum], -1, 0)
        od = (m * dpg)[None] * tab("mext")
        tau += od; ts += od * tab("ssa"); tsg += od * tab("ssa") * tab("g")
    return tau, ts / np.maximum(tau, eps), tsg / np.maximum(ts, eps)


def write_netcdf_classic(path, dims, variables, version=1, record_dim=None, numrecs=0):
    """A classic NetCDF file (CDF-1 / CDF-2) written by hand for the reader test: dims {name: length}, variables
    {name: (array, [dim names])} with dtypes int8 / S1 (char) / int16 / int32 / float32 / float64; every variable gets a
    `units` text attribute and the file one global attribute (the reader has to skip both)."""
    import struct
    TYPES = {"i1": 1, "S1": 2, "i2": 3, "i4": 4, "f4": 5, "f8": 6}
    def pad(b): return b + b"\0"*((4 - len(b) % 4) % 4)
    def name(s): return struct.pack(">I", len(s)) + pad(s.encode())
    def att(n, text): return name(n) + struct.pack(">II", 2, len(text)) + pad(text.encode())
    dnames = list(dims)
    head = b"CDF" + bytes([version]) + struct.pack(">I", numrecs)
    head += struct.pack(">II", 0x0A, len(dims))
    for d in dnames:
        head += name(d) + struct.pack(">I", 0 if d == record_dim else dims[d])
    head += struct.pack(">II", 0x0C, 1) + att("title", "written by tests/cases.py")
    infos = []
    for n, (arr, dn) in variables.items():
        arr = np.asarray(arr)
        code = arr.dtype.kind + str(arr.dtype.itemsize) if arr.dtype.kind != "S" else "S1"
        rec = bool(dn) and dn[0] == record_dim
        per = int(np.prod([dims[d] for d in dn[1:]])) if rec else int(arr.size)
        vsize = (per*arr.dtype.itemsize + 3)//4*4
        infos.append(dict(name=n, arr=arr, dn=dn, type=TYPES[code], rec=rec, per=per, vsize=vsize))
    nrec = sum(1 for i in infos if i["rec"])
    if nrec == 1:                                         # single record variable: records are not padded
        for i in infos:
            if i["rec"]: i["rsize"] = i["per"]*i["arr"].dtype.itemsize
    def var_header(i, begin):
        h = name(i["name"]) + struct.pack(">I", len(i["dn"])) + b"".join(struct.pack(">I", dnames.index(d)) for d in i["dn"])
        h += struct.pack(">II", 0x0C, 1) + att("units", "1e-6")
        h += struct.pack(">II", i["type"], i["vsize"]) + (struct.pack(">I", begin) if version == 1 else struct.pack(">Q", begin))
        return h
    hlen = len(head) + 8 + sum(len(var_header(i, 0)) for i in infos)
    pos = hlen
    for i in infos:
        if not i["rec"]: i["begin"] = pos; pos += i["vsize"]
    recsize = sum(i["vsize"] for i in infos if i["rec"]) if nrec != 1 else [i["rsize"] for i in infos if i["rec"]][0]
    for i in infos:
        if i["rec"]: i["begin"] = pos; pos += i["vsize"] if nrec != 1 else 0
    out = bytearray(head + struct.pack(">II", 0x0B, len(infos)) + b"".join(var_header(i, i["begin"]) for i in infos))
    assert len(out) == hlen
    end = max([i["begin"] + i["vsize"] for i in infos if not i["rec"]] + [hlen])
    total = end + (numrecs*recsize if nrec else 0)
    out += b"\0"*(total - len(out))
    for i in infos:
        be = i["arr"].astype(i["arr"].dtype.newbyteorder(">")).tobytes()
        if not i["rec"]:
            out[i["begin"]:i["begin"]+len(be)] = be
        else:
            rb = i["per"]*i["arr"].dtype.itemsize
            for r in range(numrecs):
                o = i["begin"] + r*recsize
                out[o:o+rb] = be[r*rb:(r+1)*rb]
    with open(path, "wb") as f:
        f.write(bytes(out))
