#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY. Generates tests/golden/*.npz by running the reference's own kernel text
(oracle/_ref, built by `make -C oracle ref` from /root/reference/src_kernels_cuda/*.cu, executed on the host).

Run in the build container only (it needs /root/reference):   python oracle/make_golden.py
The committed fixtures are DATA: seeded inputs + the outputs of the reference kernels. The k-distribution is
regenerated from its seed by the tests (rte-rrtmgp-cpp_amd/synthetic.py) and verified against the stored digest.

Cases: chained gas optics -> Planck -> LW solver -> sum, and
gas optics + Rayleigh -> combine -> SW solver -> sum, both vertical orientations, fp64 and fp32; plus
random-input solver cases in the style of tuning_kernels_cuda/{lw_solver_noscat,sw_source_adding_kernel}.py and the
element-wise optical-props kernels.
"""
import hashlib
import os
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_py as O                                            # noqa: E402
from rte_rrtmgp_cpp_amd import synthetic, pipeline               # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
MINI = dict(ngpt=32, nbnd=2, npres=10, nflav=3, nminor_lower=5, nminor_upper=3)


def kdist_digest(kd):
    h = hashlib.sha256()
    for k in sorted(kd.__dict__):
        v = kd.__dict__[k]
        if isinstance(v, np.ndarray):
            h.update(k.encode()); h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


def chained_case(dtype, top_at_1, ncol, nlay):
    ref = O.CpuKernels("ref", dtype)
    out = {}
    atm = synthetic.make_atmosphere(ncol, nlay, nbnd_lw=2, nbnd_sw=2, top_at_1=top_at_1, seed=7).astype(dtype)
    # host-side arithmetic of the reference's classes, in numpy (inputs of the kernels, stored in the fixture)
    plev = atm.p_lev.astype(np.float64)
    h2o = atm.vmr["h2o"].astype(np.float64)
    m_air = (0.028964 + 0.018016*h2o) / (1. + h2o)
    col_dry = (10.*np.abs(plev[:-1] - plev[1:])*6.02214076e23 / (1000.*m_air*100.*9.80665) / (1. + h2o)).astype(dtype)
    for kind in ("lw", "sw"):
        kd = synthetic.make_kdist(kind, **MINI)
        out[f"{kind}_kdist_digest"] = np.array(kdist_digest(kd))
        kd = ref.upload_kdist(kd)
        col_gas = ref.fill_gases(kd, atm.vmr, col_dry)
        it = ref.interpolation(kd, atm.p_lay, atm.t_lay, col_gas)
        if kind == "lw":
            out.update(p_lay=atm.p_lay, p_lev=atm.p_lev, t_lay=atm.t_lay, t_lev=atm.t_lev, t_sfc=atm.t_sfc,
                       col_dry=col_dry, col_gas=col_gas, **{f"vmr_{n}": v for n, v in atm.vmr.items()})
            out.update({f"lw_it_{k}": v for k, v in it.items()})
            tau = ref.zeros((kd.ngpt, nlay, ncol))
            ref.compute_tau_absorption(kd, it, atm.p_lay, atm.t_lay, col_gas, tau)
            src = ref.compute_planck_source(kd, it, atm.t_lay, atm.t_lev, atm.t_sfc, nlay if top_at_1 else 1)
            emis = np.ascontiguousarray(np.repeat(atm.emis_sfc.T, kd.ngpt // kd.nbnd, axis=0))
            sec = ref.lw_secants_array(ncol, kd.ngpt, 1, 4, ref.asarray(pipeline.GAUSS_DS))
            fl = ref.lw_solver_noscat(top_at_1, sec, ref.asarray(np.array([1.0])), tau, src["lay_src"], src["lev_src"],
                                      emis, src["sfc_src"])
            out.update(lw_tau=tau, lw_lay_src=src["lay_src"], lw_lev_src=src["lev_src"], lw_sfc_src=src["sfc_src"],
                       lw_sfc_src_jac=src["sfc_src_jac"], lw_sfc_emis_gpt=emis, lw_secants=sec,
                       lw_gpt_flux_up=fl["flux_up"], lw_gpt_flux_dn=fl["flux_dn"],
                       lw_flux_up=ref.sum_broadband(fl["flux_up"]), lw_flux_dn=ref.sum_broadband(fl["flux_dn"]))
            out["lw_flux_net"] = ref.net_broadband_precalc(out["lw_flux_dn"], out["lw_flux_up"])
        else:
            out.update({f"sw_it_{k}": v for k, v in it.items()})
            tau_abs = ref.zeros((kd.ngpt, nlay, ncol))
            ref.compute_tau_absorption(kd, it, atm.p_lay, atm.t_lay, col_gas, tau_abs)
            tau_ray = ref.compute_tau_rayleigh(kd, it, col_dry, col_gas)
            tau, ssa, g = ref.combine_abs_and_rayleigh(tau_abs, tau_ray)
            toa = (kd.solar_source[:, None] * atm.tsi_scaling[None, :]).astype(dtype)
            adir = np.ascontiguousarray(np.repeat(atm.sfc_alb_dir.T, kd.ngpt // kd.nbnd, axis=0))
            adif = np.ascontiguousarray(np.repeat(atm.sfc_alb_dif.T, kd.ngpt // kd.nbnd, axis=0))
            fl = ref.sw_solver_2stream(top_at_1, tau, ssa, g, atm.mu0, adir, adif, toa)
            out.update(sw_tau_abs=tau_abs, sw_tau_ray=tau_ray, sw_tau=tau, sw_ssa=ssa, sw_g=g, sw_toa_src=toa,
                       sw_alb_dir=adir, sw_alb_dif=adif, mu0=atm.mu0,
                       sw_gpt_flux_up=fl["flux_up"], sw_gpt_flux_dn=fl["flux_dn"], sw_gpt_flux_dir=fl["flux_dir"],
                       sw_flux_up=ref.sum_broadband(fl["flux_up"]), sw_flux_dn=ref.sum_broadband(fl["flux_dn"]),
                       sw_flux_dir=ref.sum_broadband(fl["flux_dir"]))
    out["meta"] = np.array([ncol, nlay, int(top_at_1), 7])
    return out


def random_solver_case(dtype, top_at_1, ncol, nlay, ngpt, seed):
    """Random optical properties (tuning_kernels_cuda/lw_solver_noscat.py:161-192 style), incl. the
    conservative-scattering and thick/thin limits that exercise the k_min / tmin / Ukkonen clamps."""
    ref = O.CpuKernels("ref", dtype)
    rng = np.random.default_rng(seed)
    shp = (ngpt, nlay, ncol)
    tau = (10.0**rng.uniform(-6, 2, shp)).astype(dtype)
    tau[0, 0, :] = 0.0                                    # tau == 0 -> series branch of `fact`
    lay = rng.uniform(5., 40., shp).astype(dtype)
    lev = rng.uniform(5., 40., (ngpt, nlay+1, ncol)).astype(dtype)
    emis = rng.uniform(0.8, 1.0, (ngpt, ncol)).astype(dtype)
    ssrc = rng.uniform(5., 40., (ngpt, ncol)).astype(dtype)
    sjac = rng.uniform(0.1, 1.0, (ngpt, ncol)).astype(dtype)
    sec = ref.lw_secants_array(ncol, ngpt, 1, 4, ref.asarray(pipeline.GAUSS_DS))
    fl = ref.lw_solver_noscat(top_at_1, sec, ref.asarray(np.array([1.0])), tau, lay, lev, emis, ssrc, sfc_src_jac=sjac)
    out = dict(lw_tau=tau, lw_lay_src=lay, lw_lev_src=lev, lw_emis=emis, lw_sfc_src=ssrc, lw_sfc_src_jac=sjac,
               lw_flux_up=fl["flux_up"], lw_flux_dn=fl["flux_dn"], lw_flux_up_jac=fl["flux_up_jac"])

    ssa = rng.uniform(0., 1., shp).astype(dtype)
    ssa[1, :, :] = 1.0                                    # conservative scattering -> k_min clamp
    ssa[2, :, :] = 0.0
    g = rng.uniform(-0.3, 0.9, shp).astype(dtype)
    mu0 = rng.uniform(0.05, 1.0, ncol).astype(dtype)
    adir_col = rng.uniform(0., 0.6, ncol).astype(dtype)   # band-uniform direct albedo
    adir = np.ascontiguousarray(np.repeat(adir_col[None, :], ngpt, axis=0))
    adif = rng.uniform(0., 0.6, (ngpt, ncol)).astype(dtype)
    inc = rng.uniform(0., 5., (ngpt, ncol)).astype(dtype)
    inc_dif = rng.uniform(0., 1., (ngpt, ncol)).astype(dtype)
    fs = ref.sw_solver_2stream(top_at_1, tau, ssa, g, mu0, adir, adif, inc)
    fd = ref.sw_solver_2stream(top_at_1, tau, ssa, g, mu0, adir, adif, inc, inc_dif)
    out.update(sw_ssa=ssa, sw_g=g, mu0=mu0, sw_alb_dir=adir, sw_alb_dif=adif, sw_inc_dir=inc, sw_inc_dif=inc_dif,
               sw_flux_up=fs["flux_up"], sw_flux_dn=fs["flux_dn"], sw_flux_dir=fs["flux_dir"],
               sw_dif_flux_up=fd["flux_up"], sw_dif_flux_dn=fd["flux_dn"])

    # element-wise optical-props kernels
    t1, w1, g1 = tau.copy(), ssa.copy(), g.copy()
    t2 = (10.0**rng.uniform(-4, 1, shp)).astype(dtype); w2 = rng.uniform(0., 1., shp).astype(dtype); g2 = rng.uniform(0., 0.9, shp).astype(dtype)
    ref.increment_2stream_by_2stream(t1, w1, g1, t2, w2, g2)
    out.update(op_t2=t2, op_w2=w2, op_g2=g2, op_inc2_tau=t1, op_inc2_ssa=w1, op_inc2_g=g1)
    t1 = tau.copy(); ref.increment_1scalar_by_1scalar(t1, t2); out["op_inc1_tau"] = t1
    nb = 2
    lims = np.array([[1, ngpt//2], [ngpt//2+1, ngpt]], dtype=np.int32)
    tb = (10.0**rng.uniform(-3, 1, (nb, nlay, ncol))).astype(dtype); wb = rng.uniform(0., 1., (nb, nlay, ncol)).astype(dtype); gb = rng.uniform(0., 0.9, (nb, nlay, ncol)).astype(dtype)
    t1, w1, g1 = tau.copy(), ssa.copy(), g.copy()
    ref.inc_2stream_by_2stream_bybnd(t1, w1, g1, tb, wb, gb, lims)
    out.update(op_lims=lims, op_tb=tb, op_wb=wb, op_gb=gb, op_incb2_tau=t1, op_incb2_ssa=w1, op_incb2_g=g1)
    t1 = tau.copy(); ref.inc_1scalar_by_1scalar_bybnd(t1, tb, lims); out["op_incb1_tau"] = t1
    t1, w1, g1 = tau.copy(), ssa.copy(), g.copy()
    ref.delta_scale_2str_k(t1, w1, g1)
    out.update(op_ds_tau=t1, op_ds_ssa=w1, op_ds_g=g1)
    out["meta"] = np.array([ncol, nlay, int(top_at_1), seed])
    return out


def glue_case(dtype, top_at_1, seed):
    """Stand-alone launchers around the solvers: the three apply_BC overloads, the two transposes, and the LW solver
    with a non-zero incident flux (which the reference's CUDA text halves: /(2 pi w) at rte_solver_kernels.cu:160, then
    *(pi w) at :189-190 -- SURVEY Q3; the fixture pins that factor)."""
    ref = O.CpuKernels("ref", dtype)
    rng = np.random.default_rng(seed)
    ncol, nlay, ngpt = 5, 7, 6
    out = {}
    base = rng.uniform(1., 2., (ngpt, nlay+1, ncol)).astype(dtype)
    inc = rng.uniform(0., 5., (ngpt, ncol)).astype(dtype)
    fac = rng.uniform(0.1, 1., ncol).astype(dtype)
    out.update(bc_base=base, bc_inc=inc, bc_factor=fac,
               bc_0=ref.apply_BC(nlay, top_at_1, base.copy()),
               bc_gpt=ref.apply_BC(nlay, top_at_1, base.copy(), inc),
               bc_fac=ref.apply_BC(nlay, top_at_1, base.copy(), inc, fac))
    a3 = rng.uniform(-1., 1., (3, 4, 5)).astype(dtype); a2 = rng.uniform(-1., 1., (7, 3)).astype(dtype)
    out.update(ro_a3=a3, ro_a2=a2, ro_321=ref.reorder123x321(a3), ro_21=ref.reorder12x21(a2))
    shp = (ngpt, nlay, ncol)
    tau = (10.0**rng.uniform(-3, 1, shp)).astype(dtype)
    lay = rng.uniform(5., 40., shp).astype(dtype); lev = rng.uniform(5., 40., (ngpt, nlay+1, ncol)).astype(dtype)
    emis = rng.uniform(0.8, 1.0, (ngpt, ncol)).astype(dtype); ssrc = rng.uniform(5., 40., (ngpt, ncol)).astype(dtype)
    sec = ref.lw_secants_array(ncol, ngpt, 1, 4, ref.asarray(pipeline.GAUSS_DS))
    fl = ref.lw_solver_noscat(top_at_1, sec, ref.asarray(np.array([1.0])), tau, lay, lev, emis, ssrc, inc_flux=inc)
    out.update(lw_tau=tau, lw_lay_src=lay, lw_lev_src=lev, lw_emis=emis, lw_sfc_src=ssrc, lw_inc=inc,
               lw_inc_flux_up=fl["flux_up"], lw_inc_flux_dn=fl["flux_dn"])
    out["meta"] = np.array([ncol, nlay, int(top_at_1), seed])
    return out


def tall_solver_case(top_at_1, nlay, seed, ncol=17, ngpt=3):
    """random-input LW / SW solver runs of the reference kernel text at the layer counts the production
    tilings serve (60 and 140 layers), 17 columns (16 column-lanes + 1). The inputs are regenerated from the seed by the tests
    (tests/cases.py:tall_inputs, digest stored); the fixture holds the reference outputs. The g = 0 run (the clear-sky form of
    the fused SW kernel) is stored as broadband sums."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    import cases
    ref = O.CpuKernels("ref", np.float64)
    I = cases.tall_inputs(seed, ncol, nlay, ngpt)
    sec = ref.lw_secants_array(ncol, ngpt, 1, 4, ref.asarray(pipeline.GAUSS_DS))
    fl = ref.lw_solver_noscat(top_at_1, sec, ref.asarray(np.array([1.0])), I["tau"], I["lay"], I["lev"], I["emis"], I["ssrc"])
    fs = ref.sw_solver_2stream(top_at_1, I["tau"], I["ssa"], I["g"], I["mu0"], I["adir"], I["adif"], I["inc"])
    f0 = ref.sw_solver_2stream(top_at_1, I["tau"], I["ssa"], np.zeros_like(I["g"]), I["mu0"], I["adir"], I["adif"], I["inc"])
    return dict(inputs_digest=np.array(cases.arrays_digest(I)),
                lw_flux_up=fl["flux_up"], lw_flux_dn=fl["flux_dn"],
                sw_flux_up=fs["flux_up"], sw_flux_dn=fs["flux_dn"], sw_flux_dir=fs["flux_dir"],
                sw_g0_bb_up=ref.sum_broadband(f0["flux_up"]), sw_g0_bb_dn=ref.sum_broadband(f0["flux_dn"]),
                sw_g0_bb_dir=ref.sum_broadband(f0["flux_dir"]),
                meta=np.array([ncol, nlay, int(top_at_1), seed, ngpt]))


def slim_chain(case):
    """Atmosphere in, broadband fluxes out: what a whole-chain replay needs (the product chain keeps no intermediates)."""
    keep = ("p_lay", "p_lev", "t_lay", "t_lev", "t_sfc", "col_dry", "mu0", "meta", "lw_kdist_digest", "sw_kdist_digest",
            "lw_flux_up", "lw_flux_dn", "lw_flux_net", "sw_flux_up", "sw_flux_dn", "sw_flux_dir", "sw_toa_src",
            "lw_sfc_emis_gpt", "sw_alb_dir", "sw_alb_dif")
    return {k: v for k, v in case.items() if k in keep or k.startswith("vmr_")}


def optics_lut_digest(lut):
    h = hashlib.sha256()
    for k in sorted(lut):
        h.update(k.encode()); h.update(np.ascontiguousarray(np.asarray(lut[k], dtype=np.float64)).tobytes())
    return h.hexdigest()


def cloud_case(dtype, kind, seed):
    """The reference's own CPU class Cloud_optics (src/Cloud_optics.cpp:29-232, compiled unmodified: oracle/refcpu_runner.cpp)
    on a synthetic LUT: both cloud_optics() overloads, particle sizes over the whole table incl. both end points, cloud-free
    cells, cells with only liquid / only ice."""
    import refcpu_py
    rng = np.random.default_rng(seed)
    nbnd, nlay, ncol = 6, 9, 11
    lut = synthetic.make_cloud_lut(nbnd, kind)
    shp = (nlay, ncol)
    lwp = np.where(rng.uniform(size=shp) < 0.35, 0., rng.uniform(0., 60., shp))
    iwp = np.where(rng.uniform(size=shp) < 0.35, 0., rng.uniform(0., 40., shp))
    rel = rng.uniform(lut["radliq_lwr"], lut["radliq_upr"], shp); dei = rng.uniform(lut["diamice_lwr"], lut["diamice_upr"], shp)
    rel[0, 0], rel[0, 1] = lut["radliq_lwr"], lut["radliq_upr"]
    dei[0, 0], dei[0, 1] = lut["diamice_lwr"], lut["diamice_upr"]
    lwp[0, :2] = 30.; iwp[0, :2] = 20.
    lutd = {k: (v.astype(dtype) if isinstance(v, np.ndarray) else v) for k, v in lut.items()}
    ins = [a.astype(dtype) for a in (lwp, iwp, rel, dei)]
    t2, w2, g2, t1 = refcpu_py.cloud_optics(dtype, lutd, *ins)
    return dict(lut_kind=np.array(kind), lut_nbnd=np.array(nbnd), lut_digest=np.array(optics_lut_digest(lut)),
                clwp=ins[0], ciwp=ins[1], reliq=ins[2], deice=ins[3], tau_2str=t2, ssa_2str=w2, g_2str=g2, tau_1scl=t1,
                meta=np.array([ncol, nlay, 0, seed]))


def aerosol_case(dtype, table, seed):
    """The reference's own CPU class Aerosol_optics (src/Aerosol_optics.cpp:24-224, compiled unmodified) on the real CAMS
    tables of the reference tree (data/aerosol_optics.nc = tests/golden/aerosol_optics.nc) and on a synthetic table with
    another band count. Surface-first columns (the CPU text has no abs() on dp) and rh <= 1 (its humidity-class search is
    unbounded above the last class); two of the eleven species are given as profiles, as in the all-sky input file."""
    import refcpu_py
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    import cases
    with tempfile.TemporaryDirectory() as d:
        lut = cases.real_aerosol_lut(d) if table == "real" else synthetic.make_aerosol_lut(5)
    atm = synthetic.make_atmosphere(13, 21, aerosols=True, top_at_1=False, seed=seed)
    rh = np.minimum(atm.rh, 1.0)
    rh[0, 0] = lut["rh_upper"][3]                       # exactly on a class boundary
    aermr = [atm.aermr["aermr%02d" % i].astype(dtype) for i in range(1, 12)]
    lutd = {k: v.astype(dtype) for k, v in lut.items()}
    tau, ssa, g = refcpu_py.aerosol_optics(dtype, lutd, aermr, rh.astype(dtype), atm.p_lev.astype(dtype))
    out = dict(table=np.array(table), lut_digest=np.array(optics_lut_digest(lut)), rh=rh.astype(dtype), p_lev=atm.p_lev.astype(dtype),
               tau=tau, ssa=ssa, g=g, meta=np.array([13, 21, 0, seed]))
    out.update({"aermr%02d" % (i+1): m for i, m in enumerate(aermr)})
    return out


def main():
    O.build(ref=True)
    subprocess.run(["make", "-C", HERE, "refcpu"], check=True, stdout=subprocess.DEVNULL)
    os.makedirs(GOLDEN, exist_ok=True)
    total = 0
    for dtype, tag in ((np.float64, "f64"), (np.float32, "f32")):
        for top in (False, True):
            ncol, nlay = (3, 30) if dtype == np.float64 else (2, 12)
            for name, case in (
                    (f"chain_{tag}_top{int(top)}", chained_case(dtype, top, ncol, nlay)),
                    (f"random_{tag}_top{int(top)}", random_solver_case(dtype, top, 5, 19, 8, 11 + int(top)))):
                path = os.path.join(GOLDEN, name + ".npz")
                np.savez_compressed(path, **case)
                total += os.path.getsize(path)
                print(f"wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)")
    if True:
        for dtype, tag in ((np.float64, "f64"), (np.float32, "f32")):
            for top in (False, True):
                path = os.path.join(GOLDEN, f"glue_{tag}_top{int(top)}.npz")
                np.savez_compressed(path, **glue_case(dtype, top, 21 + int(top)))
                total += os.path.getsize(path)
                print(f"wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)")
    # production tilings: 60- and 140-layer solver runs and a 140-layer whole chain
    for top in (False, True):
        for nlay in (60, 140):
            path = os.path.join(GOLDEN, f"tall_f64_top{int(top)}_nlay{nlay}.npz")
            np.savez_compressed(path, **tall_solver_case(top, nlay, 50 + nlay + int(top)))
            total += os.path.getsize(path)
            print(f"wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)")
        path = os.path.join(GOLDEN, f"chainbb_f64_top{int(top)}_17x140.npz")
        np.savez_compressed(path, **slim_chain(chained_case(np.float64, top, 17, 140)))
        total += os.path.getsize(path)
        print(f"wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)")
    # the reference's CPU classes for cloud / aerosol optics
    for dtype, tag in ((np.float64, "f64"), (np.float32, "f32")):
        cases_ = [(f"cloud_{tag}_lw", cloud_case(dtype, "lw", 31)), (f"cloud_{tag}_sw", cloud_case(dtype, "sw", 32)),
                  (f"aerosol_{tag}_real", aerosol_case(dtype, "real", 41)), (f"aerosol_{tag}_synthetic", aerosol_case(dtype, "synthetic", 42))]
        for name, case in cases_:
            path = os.path.join(GOLDEN, name + ".npz")
            np.savez_compressed(path, **case)
            total += os.path.getsize(path)
            print(f"wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)")
    # degenerate shape: 1 column x 4 layers
    case = chained_case(np.float64, False, 1, 4)
    path = os.path.join(GOLDEN, "chain_f64_top0_1x4.npz")
    np.savez_compressed(path, **case)
    total += os.path.getsize(path)
    print(f"total {total/1024:.0f} KiB")


if __name__ == "__main__":
    main()
