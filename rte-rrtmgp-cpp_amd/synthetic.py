"""Seeded synthetic inputs for the RTE+RRTMGP hot path.

The reference ships neither its k-distribution files nor any atmosphere (rrtmgp-data submodule is empty,
SURVEY F2), so every test and benchmark runs on:

* ``make_kdist``  -- a synthetic k-distribution with the REAL shapes of rrtmgp-gas-lw-g256 /
  the tuning scripts (tuning_kernels_cuda/compute_tau_absorption_kernel.py:174-189: ntemp 14, neta 9,
  npres 59, nflav 10, ngas 7, 16 g-points per band, 44/19 minor intervals, nPlanckTemp 196), already in the
  layouts the kernels see, i.e. AFTER Gas_optics_rrtmgp::init_abs_coeffs
  (/root/reference/src/Gas_optics_rrtmgp.cpp:539-742): kmajor(ntemp,neta,npres+1,ngpt), krayl(ntemp,neta,ngpt,2),
  kminor(ntemp,neta,nminork), 1-based index arrays.
* ``make_atmosphere`` -- the RCEMIP analytic column (/root/reference/rcemip/test_rcemip_input.py:20-75) on uniform
  layers to 70 km, surface first, with a seeded per-column perturbation; optional all-sky clouds
  (/root/reference/allsky/allsky_init.py:167-176).

Array convention: numpy arrays are C-ordered with REVERSED dimensions, so that their memory is the
column-major, column-fastest layout of the reference: Fortran (ncol,nlay,ngpt)  <->  numpy shape (ngpt,nlay,ncol).
"""
from dataclasses import dataclass, field
import os
import numpy as np

GAS_NAMES = ["h2o", "co2", "o3", "n2o", "ch4", "o2", "n2"]      # 1-based gas indices 1..7 in col_gas


@dataclass
class KDist:
    """k-distribution in kernel layout. Shapes are numpy (reversed Fortran) shapes."""
    kind: str                      # "lw" or "sw"
    ngpt: int
    nbnd: int
    ngas: int
    nflav: int
    neta: int
    npres: int
    ntemp: int
    gas_names: list
    idx_h2o: int
    flavor: np.ndarray             # (nflav, 2)            int32, gas indices 0..ngas
    gpoint_flavor: np.ndarray      # (ngpt, 2)             int32, 1-based
    band_lims_gpt: np.ndarray      # (nbnd, 2)             int32, 1-based inclusive
    gpoint_bands: np.ndarray       # (ngpt,)               int32, 1-based
    press_ref_log: np.ndarray      # (npres,)
    temp_ref: np.ndarray           # (ntemp,)
    press_ref_log_delta: float
    temp_ref_min: float
    temp_ref_delta: float
    press_ref_trop_log: float
    vmr_ref: np.ndarray            # (ntemp, ngas+1, 2)
    kmajor: np.ndarray             # (ngpt, npres+1, neta, ntemp)
    kminor_lower: np.ndarray       # (nminorklower, neta, ntemp)
    kminor_upper: np.ndarray
    minor_limits_gpt_lower: np.ndarray   # (nminor, 2) int32
    minor_limits_gpt_upper: np.ndarray
    minor_scales_with_density_lower: np.ndarray   # (nminor,) int8
    minor_scales_with_density_upper: np.ndarray
    scale_by_complement_lower: np.ndarray         # (nminor,) int8
    scale_by_complement_upper: np.ndarray
    idx_minor_lower: np.ndarray    # (nminor,) int32, gas index 1..ngas
    idx_minor_upper: np.ndarray
    idx_minor_scaling_lower: np.ndarray   # (nminor,) int32, 0 = none
    idx_minor_scaling_upper: np.ndarray
    kminor_start_lower: np.ndarray  # (nminor,) int32, 1-based
    kminor_start_upper: np.ndarray
    # LW only
    nPlanckTemp: int = 0
    totplnk_delta: float = 0.0
    totplnk: np.ndarray = None      # (nbnd, nPlanckTemp)
    planck_frac: np.ndarray = None  # (ngpt, npres+1, neta, ntemp)
    # SW only
    krayl: np.ndarray = None        # (2, ngpt, neta, ntemp)
    solar_source: np.ndarray = None # (ngpt,)

    def astype(self, dtype):
        """Copy with all floating arrays cast to ``dtype`` (float32 build = RTE_USE_SP)."""
        out = {}
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray) and v.dtype.kind == "f":
                out[k] = np.ascontiguousarray(v.astype(dtype))
            else:
                out[k] = v
        return KDist(**out)


def _planck_band_integrals(temps, band_lims_wn):
    """Band-integrated Planck radiance [W m-2 sr-1] for wavenumber bands [cm-1]."""
    h, c, kb = 6.62607015e-34, 2.99792458e8, 1.380649e-23
    out = np.zeros((len(band_lims_wn), len(temps)))
    for ib, (w0, w1) in enumerate(band_lims_wn):
        nu = np.linspace(w0, w1, 400) * 100.0            # m-1
        for it, T in enumerate(temps):
            B = 2.0*h*c*c*nu**3 / np.expm1(h*c*nu/(kb*T))
            out[ib, it] = np.trapezoid(B, nu)
    return out


def make_kdist(kind="lw", ngpt=256, nbnd=16, seed=1234, ntemp=14, npres=59, neta=9, nflav=10,
               nminor_lower=44, nminor_upper=19, nPlanckTemp=196):
    """Synthetic k-distribution (see module docstring). Deterministic for a given argument tuple."""
    assert ngpt % nbnd == 0
    rng = np.random.default_rng(seed + (0 if kind == "lw" else 1))
    gpb = ngpt // nbnd
    ngas = len(GAS_NAMES)

    # --- reference grids (values: SURVEY section 8(d); lengths are what matters) ---
    temp_ref = 160.0 + 15.0*np.arange(ntemp)
    press_ref = np.exp(np.linspace(np.log(109663.31), np.log(1.005183574463), npres))
    press_ref_log = np.log(press_ref)
    press_ref_log_delta = (press_ref_log[-1] - press_ref_log[0]) / (npres - 1)
    temp_ref_delta = (temp_ref[-1] - temp_ref[0]) / (ntemp - 1)
    press_ref_trop = 9948.431564193395

    # --- flavors: pairs of key species (0 = dry air) ---
    pairs = [(1, 2), (1, 3), (2, 3), (1, 4), (1, 5), (2, 4), (2, 2), (1, 1), (3, 3), (6, 6), (1, 6), (2, 6)]
    flavor = np.array(pairs[:nflav], dtype=np.int32)
    low = rng.integers(1, nflav+1, size=nbnd)
    upp = rng.integers(1, nflav+1, size=nbnd)
    gpoint_flavor = np.zeros((ngpt, 2), dtype=np.int32)
    gpoint_flavor[:, 0] = np.repeat(low, gpb)
    gpoint_flavor[:, 1] = np.repeat(upp, gpb)
    band_lims_gpt = np.stack([1 + gpb*np.arange(nbnd), gpb*(1 + np.arange(nbnd))], axis=1).astype(np.int32)
    gpoint_bands = np.repeat(1 + np.arange(nbnd), gpb).astype(np.int32)

    # --- reference volume mixing ratios vmr_ref(2, 0:ngas, ntemp) ---
    base_vmr = np.array([1.0, 5e-3, 3.6e-4, 2e-6, 3.1e-7, 1.7e-6, 0.209, 0.781])
    vmr_ref = np.zeros((ntemp, ngas+1, 2))
    for it in range(ntemp):
        f = 1.0 + 0.04*(it - ntemp/2)
        vmr_ref[it, :, 0] = base_vmr * f
        vmr_ref[it, :, 1] = base_vmr * (2.0 - f) * np.array([1, 1e-3, 1, 2.5, 1, 1, 1, 1])
    vmr_ref[:, 0, :] = 1.0

    # --- major absorption coefficients: smooth positive in (T, eta, p, g) ---
    gq = (np.arange(gpb) + 0.5) / gpb                                # position within band
    k_g = np.tile(10.0**(-25.5 + 4.5*gq**1.5), nbnd) * np.repeat(10.0**rng.uniform(-1.0, 0.7, nbnd), gpb)
    T = temp_ref[None, None, None, :]
    eta = (np.arange(neta)/(neta-1))[None, None, :, None]
    lnp = np.concatenate([press_ref_log, press_ref_log[-1:]+press_ref_log_delta])[None, :, None, None]
    g_idx = np.arange(ngpt)[:, None, None, None]
    kmajor = (k_g[:, None, None, None]
              * np.exp(0.35*(lnp - press_ref_log[0]) * (0.5 + 0.5*np.cos(0.37*g_idx)))
              * (T/250.0)**(1.0 + 0.8*np.sin(0.11*g_idx))
              * (0.6 + 0.8*eta + 0.3*eta*eta*np.cos(0.05*g_idx)))
    kmajor = np.ascontiguousarray(kmajor)

    # --- minor contributors: each interval spans one full band ---
    def make_minor(nminor):
        bands = np.sort(rng.integers(0, nbnd, size=nminor))
        limits = band_lims_gpt[bands].astype(np.int32)
        start = (1 + gpb*np.arange(nminor)).astype(np.int32)
        nk = gpb*nminor
        swd = (rng.random(nminor) < 0.6).astype(np.int8)
        sbc = (rng.random(nminor) < 0.4).astype(np.int8)
        idx_minor = rng.integers(1, ngas+1, size=nminor).astype(np.int32)
        idx_scal = np.where(rng.random(nminor) < 0.5, rng.integers(1, ngas+1, size=nminor), 0).astype(np.int32)
        # make sure every (density, complement, scaling-gas) combination occurs
        for i, (a, b, c) in enumerate([(1, 1, 2), (1, 0, 6), (1, 1, 0), (0, 0, 0), (0, 1, 3)]):
            if i < nminor:
                swd[i], sbc[i], idx_scal[i] = a, b, c
        kk = np.arange(nk)[:, None, None]
        km = (10.0**(-24.0 + 2.0*((kk % gpb)+0.5)/gpb)
              * (temp_ref[None, None, :]/250.0)**1.5
              * (0.8 + 0.4*(np.arange(neta)/(neta-1))[None, :, None])
              * (1.0 + 0.3*np.sin(0.7*kk)))
        # density-scaled contributors carry an extra 0.01*p/T factor (~1..4): keep taus comparable
        return dict(k=np.ascontiguousarray(km), limits=limits, start=start, swd=swd, sbc=sbc,
                    idx=idx_minor, idx_scal=idx_scal)

    ml = make_minor(nminor_lower)
    mu = make_minor(nminor_upper)

    kd = dict(
        kind=kind, ngpt=ngpt, nbnd=nbnd, ngas=ngas, nflav=nflav, neta=neta, npres=npres, ntemp=ntemp,
        gas_names=list(GAS_NAMES), idx_h2o=1,
        flavor=flavor, gpoint_flavor=gpoint_flavor, band_lims_gpt=band_lims_gpt, gpoint_bands=gpoint_bands,
        press_ref_log=press_ref_log, temp_ref=temp_ref,
        press_ref_log_delta=float(press_ref_log_delta), temp_ref_min=float(temp_ref[0]),
        temp_ref_delta=float(temp_ref_delta), press_ref_trop_log=float(np.log(press_ref_trop)),
        vmr_ref=np.ascontiguousarray(vmr_ref), kmajor=kmajor,
        kminor_lower=ml["k"], kminor_upper=mu["k"],
        minor_limits_gpt_lower=ml["limits"], minor_limits_gpt_upper=mu["limits"],
        minor_scales_with_density_lower=ml["swd"], minor_scales_with_density_upper=mu["swd"],
        scale_by_complement_lower=ml["sbc"], scale_by_complement_upper=mu["sbc"],
        idx_minor_lower=ml["idx"], idx_minor_upper=mu["idx"],
        idx_minor_scaling_lower=ml["idx_scal"], idx_minor_scaling_upper=mu["idx_scal"],
        kminor_start_lower=ml["start"], kminor_start_upper=mu["start"],
    )

    if kind == "lw":
        band_wn = np.linspace(10.0, 3250.0, nbnd+1)
        band_lims_wn = np.stack([band_wn[:-1], band_wn[1:]], axis=1)
        Tpl = 160.0 + np.arange(nPlanckTemp) * (temp_ref[-1]-temp_ref[0])/(nPlanckTemp-1)
        totplnk = _planck_band_integrals(Tpl, band_lims_wn)            # (nbnd, nPlanckTemp)
        pf = 0.2 + rng.random((ngpt, 1, 1, 1)) + 0.15*np.sin(0.3*g_idx + 0.02*T) \
            + 0.1*eta + 0.05*np.cos(0.2*lnp)
        pf = np.broadcast_to(pf, kmajor.shape).copy()
        pf = pf.reshape(nbnd, gpb, npres+1, neta, ntemp)
        pf /= pf.sum(axis=1, keepdims=True)
        kd.update(nPlanckTemp=nPlanckTemp, totplnk_delta=float((temp_ref[-1]-temp_ref[0])/(nPlanckTemp-1)),
                  totplnk=np.ascontiguousarray(totplnk),
                  planck_frac=np.ascontiguousarray(pf.reshape(ngpt, npres+1, neta, ntemp)))
    else:
        gi = np.arange(ngpt)[None, :, None, None]
        krayl = (10.0**(-27.0 + 1.5*(gi/ngpt))
                 * (1.0 + 0.1*(np.arange(neta)/(neta-1))[None, None, :, None])
                 * (1.0 + 0.002*(temp_ref[None, None, None, :]-250.0))
                 * np.array([1.0, 0.9])[:, None, None, None])
        ss = 0.5 + rng.random(ngpt)
        ss *= 1360.85 / ss.sum()
        kd.update(krayl=np.ascontiguousarray(krayl), solar_source=ss)

    return KDist(**kd)


@dataclass
class Atmosphere:
    ncol: int
    nlay: int
    top_at_1: bool
    p_lay: np.ndarray      # (nlay, ncol)
    p_lev: np.ndarray      # (nlay+1, ncol)
    t_lay: np.ndarray
    t_lev: np.ndarray
    t_sfc: np.ndarray      # (ncol,)
    vmr: dict              # gas name -> (nlay, ncol) array
    emis_sfc: np.ndarray   # (ncol, nbnd)   [Fortran (nbnd, ncol)]
    sfc_alb_dir: np.ndarray
    sfc_alb_dif: np.ndarray
    mu0: np.ndarray        # (ncol,)
    tsi_scaling: np.ndarray
    lwp: np.ndarray = None
    iwp: np.ndarray = None
    rel: np.ndarray = None
    dei: np.ndarray = None
    rh: np.ndarray = None  # (nlay, ncol) relative humidity [0..1], aerosol optics only
    aermr: dict = None     # "aermr01".."aermr11" -> (nlay, ncol) mixing ratio [kg/kg], or an (nlay,) profile

    def astype(self, dtype):
        out = {}
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray) and v.dtype.kind == "f":
                out[k] = np.ascontiguousarray(v.astype(dtype))
            elif isinstance(v, dict):
                out[k] = {n: np.ascontiguousarray(a.astype(dtype)) for n, a in v.items()}
            else:
                out[k] = v
        return Atmosphere(**out)


def _rcemip_profile(z):
    """/root/reference/rcemip/test_rcemip_input.py:20-54."""
    q_0, z_q1, z_q2, z_t = 0.01864, 4.0e3, 7.5e3, 15.e3
    q = q_0 * np.exp(-z/z_q1) * np.exp(-(z/z_q2)**2)
    q_t = q_0 * np.exp(-z_t/z_q1) * np.exp(-(z_t/z_q2)**2)
    above = z > z_t
    q = np.where(above, q_t, q)
    T_0, gamma = 300., 6.7e-3
    Tv_0 = (1. + 0.608*q_0)*T_0
    Tv_t = Tv_0 - gamma*z_t
    Tv = np.where(above, Tv_t, Tv_0 - gamma*z)
    T = Tv / (1. + 0.608*q)
    g, Rd, p0 = 9.79764, 287.04, 101480.
    p = p0 * (Tv / Tv_0)**(g/(Rd*gamma))
    p_tmp = p0 * (Tv_t/Tv_0)**(g/(Rd*gamma)) * np.exp(-((g*(z-z_t)) / (Rd*Tv_t)))
    p = np.where(above, p_tmp, p)
    return p, q, T


def make_atmosphere(ncol, nlay=140, nbnd_lw=16, nbnd_sw=16, seed=1234, top_at_1=False, clouds=False, z_top=70.e3, aerosols=False,
                    col_range=None):
    """RCEMIP analytic column replicated over ``ncol`` columns with a seeded +-1 K / +-5 % humidity perturbation.
    col_range = (start, stop): only those columns of the ncol-column atmosphere are built (a rank's share of a sharded job:
    column c is the same column whatever the range), without ever holding the whole domain."""
    if col_range is not None:
        if aerosols:
            raise ValueError("col_range is for the benchmark atmospheres (clear sky, clouds)")
        s_, e_ = col_range
        rng = np.random.default_rng(seed)
        dT_all = rng.uniform(-1.0, 1.0, size=ncol); dq_all = rng.uniform(0.95, 1.05, size=ncol)
        return _make_atmosphere(e_ - s_, nlay, nbnd_lw, nbnd_sw, seed, top_at_1, clouds, z_top, False,
                                _perturbation=(dT_all[s_:e_], dq_all[s_:e_]), _col0=s_)
    return _make_atmosphere(ncol, nlay, nbnd_lw, nbnd_sw, seed, top_at_1, clouds, z_top, aerosols)


def _make_atmosphere(ncol, nlay, nbnd_lw, nbnd_sw, seed, top_at_1, clouds, z_top, aerosols, _perturbation=None, _col0=0):
    rng = np.random.default_rng(seed)
    dz = z_top / nlay
    z = dz/2 + dz*np.arange(nlay)
    zh = dz*np.arange(nlay+1)
    p_lay, q, T_lay = _rcemip_profile(z)
    p_lev, _, T_lev = _rcemip_profile(zh)
    Rd_Rv = 287.04 / 461.5
    h2o = q / (Rd_Rv * (1. - q))
    p_hpa = p_lay/100.
    o3 = np.maximum(1e-13, 3.6478 * p_hpa**0.83209 * np.exp(-p_hpa/11.3515) * 1e-6)

    dT = rng.uniform(-1.0, 1.0, size=ncol)
    dq = rng.uniform(0.95, 1.05, size=ncol)
    if _perturbation is not None:
        dT, dq = (np.array(x, dtype=np.float64) for x in _perturbation)
    if os.environ.get("RRX_SYNTH_IDENTICAL_COLUMNS"):        # diagnostic only: the reference's own RCEMIP input
        dT[:] = 0.0; dq[:] = 1.0

    def col2d(prof):
        return np.repeat(prof[:, None], ncol, axis=1)

    atm = dict(
        ncol=ncol, nlay=nlay, top_at_1=top_at_1,
        p_lay=col2d(p_lay), p_lev=col2d(p_lev),
        t_lay=col2d(T_lay) + dT[None, :], t_lev=col2d(T_lev) + dT[None, :],
        t_sfc=300.0 + dT,
        vmr=dict(
            h2o=col2d(h2o) * dq[None, :], o3=col2d(o3),
            co2=np.full((nlay, ncol), 348.e-6), ch4=np.full((nlay, ncol), 1650.e-9),
            n2o=np.full((nlay, ncol), 306.e-9), n2=np.full((nlay, ncol), 0.7808), o2=np.full((nlay, ncol), 0.2095)),
        emis_sfc=np.full((ncol, nbnd_lw), 0.98 if clouds else 1.0),
        sfc_alb_dir=np.full((ncol, nbnd_sw), 0.07), sfc_alb_dif=np.full((ncol, nbnd_sw), 0.07),
        mu0=np.full(ncol, np.cos(np.deg2rad(42.05))), tsi_scaling=np.full(ncol, 551.58/1360.85),
    )
    if clouds:
        flag = ((np.arange(1, ncol+1) + _col0) % 3 > 0)          # (global column index: a rank's share is the same columns)
        mask = (atm["p_lay"] > 1.e4) & (atm["p_lay"] < 9.e4) & flag[None, :]
        atm["lwp"] = np.where(mask & (atm["t_lay"] > 263.), 10., 0.)
        atm["iwp"] = np.where(mask & (atm["t_lay"] < 273.), 10., 0.)
        atm["rel"] = np.where(atm["lwp"] > 0., 12.0, 0.)
        atm["dei"] = np.where(atm["iwp"] > 0., 95.0, 0.)
    if aerosols:
        # CAMS-like input of the reference's aerosol test case (test_rte_rrtmgp.cu:303-320): relative humidity spanning every
        # humidity class (and a little beyond the last bound), 11 mixing ratios decaying with height; two of them given as
        # plain profiles, the form read_and_set_aer also accepts
        rng_a = np.random.default_rng(seed + 77)
        atm["rh"] = np.clip(0.05 + 1.4*np.exp(-z/6.e3)[:, None] * rng_a.uniform(0.6, 1.0, size=(nlay, ncol)), 0., 1.02)
        scale = [4e-8, 3e-8, 1e-9, 2e-9, 5e-9, 8e-9, 6e-9, 3e-9, 1e-9, 7e-10, 9e-9]
        atm["aermr"] = {}
        for i, sc in enumerate(scale, start=1):
            prof = sc * np.exp(-z/(1.5e3 + 400.*i))
            if i in (5, 10):
                atm["aermr"]["aermr%02d" % i] = prof
            else:
                atm["aermr"]["aermr%02d" % i] = prof[:, None] * rng_a.uniform(0.5, 1.5, size=(1, ncol)) * rng_a.uniform(0.9, 1.1, size=(nlay, ncol))
    if top_at_1:
        for k in ("p_lay", "p_lev", "t_lay", "t_lev", "lwp", "iwp", "rel", "dei", "rh"):
            if atm.get(k) is not None:
                atm[k] = np.ascontiguousarray(atm[k][::-1])
        atm["vmr"] = {n: np.ascontiguousarray(a[::-1]) for n, a in atm["vmr"].items()}
        if atm.get("aermr") is not None:
            atm["aermr"] = {n: np.ascontiguousarray(a[::-1]) for n, a in atm["aermr"].items()}
    for k, v in list(atm.items()):
        if isinstance(v, np.ndarray):
            atm[k] = np.ascontiguousarray(v)
    if atm.get("aermr") is not None:
        atm["aermr"] = {n: np.ascontiguousarray(a) for n, a in atm["aermr"].items()}
    return Atmosphere(**atm)


def make_aerosol_lut(nbnd, nhum=12, nphobic=14, nphilic=7):
    """Synthetic aerosol-optics tables with the shapes and value ranges of data/aerosol_optics.nc as the loader hands them to
    Aerosol_optics_gpu (/root/reference/src_test/Radiation_solver.cu:366-401): hydrophobic tables stored as numpy
    (nphobic, nbnd), hydrophilic (nphilic, nhum, nbnd), rh_upper (nhum,) = the upper bounds of the humidity classes."""
    rh_upper = np.array([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.85, 0.9, 0.95, 1.0])[:nhum].copy()
    rh_upper[-1] = 1.0
    b = np.arange(nbnd)[None, :]
    s = np.arange(nphobic)[:, None]
    lut = dict(rh_upper=rh_upper,
               mext_phobic=200. * (1. + b)**1.3 * (1. + 0.3*np.sin(1.7*s + 0.2*b)),
               ssa_phobic=np.clip(0.55 + 0.4*np.cos(0.9*s + 0.35*b)**2, 0., 0.999999),
               g_phobic=0.35 + 0.45*np.sin(0.5*s + 0.15*b)**2)
    s3 = np.arange(nphilic)[:, None, None]; h = np.arange(nhum)[None, :, None]; b3 = b[None, :, :]
    lut.update(mext_philic=120. * (1. + b3)**1.2 * (1. + 0.12*h)**2 * (1. + 0.25*np.cos(1.1*s3 + 0.3*b3)),
               ssa_philic=np.clip(0.7 + 0.29*np.sin(0.6*s3 + 0.2*h + 0.1*b3)**2, 0., 0.999999),
               g_philic=0.5 + 0.3*np.cos(0.4*s3 + 0.1*h + 0.2*b3)**2)
    return {k: np.ascontiguousarray(v) for k, v in lut.items()}


def aerosol_lut_from_file_vars(v):
    """The same dictionary from the variables of a real aerosol_optics.nc (name -> (array, dims), e.g. rrxio.read of the
    converted file)."""
    g = lambda n: np.ascontiguousarray(np.asarray(v[n][0], dtype=np.float64))
    return dict(rh_upper=g("relative_humidity2"),
                mext_phobic=g("mass_ext_sw_hydrophobic"), ssa_phobic=g("ssa_sw_hydrophobic"), g_phobic=g("asymmetry_sw_hydrophobic"),
                mext_philic=g("mass_ext_sw_hydrophilic"), ssa_philic=g("ssa_sw_hydrophilic"), g_philic=g("asymmetry_sw_hydrophilic"))


def make_cloud_lut(nbnd, kind="lw", nsize_liq=20, nsize_ice=18, seed=99):
    """Synthetic cloud-optics LUT with the shapes of rrtmgp-clouds-*.nc after roughness selection
    (/root/reference/src/Cloud_optics.cpp:29-69): lut_*(nsize, nbnd) stored as numpy (nbnd, nsize)."""
    rng = np.random.default_rng(seed + (0 if kind == "lw" else 1))
    rl = np.linspace(2.5, 21.5, nsize_liq)
    di = np.linspace(10., 180., nsize_ice)
    b = np.arange(nbnd)[:, None]
    lut = dict(
        radliq_lwr=2.5, radliq_upr=21.5, diamice_lwr=10., diamice_upr=180.,
        nsize_liq=nsize_liq, nsize_ice=nsize_ice,
        lut_extliq=1.5/rl[None, :] * (1 + 0.05*np.sin(b)),
        lut_ssaliq=np.clip((0.5 if kind == "lw" else 0.97) + 0.02*np.cos(b + rl[None, :]/5), 0, 0.999999),
        lut_asyliq=0.8 + 0.05*np.sin(b/3 + rl[None, :]/10),
        lut_extice=3.0/di[None, :] * (1 + 0.05*np.cos(b)),
        lut_ssaice=np.clip((0.45 if kind == "lw" else 0.95) + 0.03*np.sin(b + di[None, :]/40), 0, 0.999999),
        lut_asyice=0.75 + 0.08*np.cos(b/4 + di[None, :]/60),
    )
    _ = rng
    return {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in lut.items()}
