"""Launcher-level orchestration of one LW / SW solve, generic over a backend object.

Mirrors, call for call, what the reference's host classes do around the kernel launchers:
  Radiation_solver_longwave::solve_gpu   /root/reference/src_test/Radiation_solver.cu:419-680
  Radiation_solver_shortwave::solve_gpu  /root/reference/src_test/Radiation_solver.cu:683-950
  Gas_optics_rrtmgp_gpu::gas_optics / compute_gas_taus / source   /root/reference/src_cuda/Gas_optics_rrtmgp.cu:907-1201
  Rte_lw_gpu::rte_lw                     /root/reference/src_cuda/Rte_lw.cu:60-136
  Rte_sw_gpu::rte_sw                     /root/reference/src_cuda/Rte_sw.cu:116-161
  Fluxes_broadband_gpu::reduce           /root/reference/src_cuda/Fluxes.cu:39-54

The backend is either HipKernels (product path, torch CUDA tensors) or, in tests only, a CPU checker from
oracle/oracle_py.py (numpy). This module never imports the oracle.
"""
import os

import numpy as np

from ._ffi import BoolArg

# Gauss-Jacobi-5 quadrature, R. J. Hogan 2023 (values as in /root/reference/src/Rte_lw.cpp:140-152); (n_angles, point)
MAX_GAUSS_PTS = 4
GAUSS_DS = np.array([
    [1./0.6096748751, 0., 0., 0.],
    [1./0.2509907356, 1/0.7908473988, 0., 0.],
    [1./0.1024922169, 1/0.4417960320, 1./0.8633751621, 0.],
    [1./0.0454586727, 1/0.2322334416, 1./0.5740198775, 1./0.903077597]])
GAUSS_WTS = np.array([
    [1., 0., 0., 0.],
    [0.2300253764, 0.7699746236, 0., 0.],
    [0.0437820218, 0.3875796738, 0.5686383044, 0.],
    [0.0092068785, 0.1285704278, 0.4323381850, 0.4298845087]])


def upload_atmosphere(be, atm):
    from .synthetic import Atmosphere
    out = {}
    for k, v in atm.__dict__.items():
        if isinstance(v, np.ndarray):
            out[k] = be.asarray(v)
        elif isinstance(v, dict):
            out[k] = {n: be.asarray(a) for n, a in v.items()}
        else:
            out[k] = v
    return Atmosphere(**out)


def _sfc_lay(atm):
    # /root/reference/src_cuda/Gas_optics_rrtmgp.cu:1190 : play(1,1) > play(1,nlay) ? 1 : nlay
    return atm.nlay if atm.top_at_1 else 1


def gas_state(be, kd, atm, col_dry=None, interpolate=True):
    """col_dry, col_gas and the interpolation state shared by the optical-depth and source kernels. interpolate=False:
    the consumers compute the interpolation state themselves (the "direct" entry points), nothing is materialised."""
    if col_dry is None:
        col_dry = be.get_col_dry(atm.vmr["h2o"], atm.p_lev)
    col_gas = be.fill_gases(kd, atm.vmr, col_dry)
    it = be.interpolation(kd, atm.p_lay, atm.t_lay, col_gas) if interpolate else None
    return col_dry, col_gas, it


def _use_direct(be, direct):
    return hasattr(be, "gas_optics_lw_direct") if direct is None else bool(direct)


def solve_lw(be, kd, atm, col_dry=None, cloud_lut=None, n_gauss_angles=1, do_broadband=False, keep=False, direct=None, lite=None,
             fuse_clouds=True):
    """lite (default: broadband mode with one angle on a backend that has it): Planck fractions + band Planck functions, the
    sources are formed inside the broadband solver (the "Planck-lite" chain); keep=True materialises them for the caller."""
    ncol, nlay, ngpt = atm.ncol, atm.nlay, kd.ngpt
    direct = _use_direct(be, direct)
    if lite is None:
        lite = direct and do_broadband and n_gauss_angles == 1 and hasattr(be, "planck_fractions")
    col_dry, col_gas, it = gas_state(be, kd, atm, col_dry, interpolate=not direct)

    fr = None
    # all-sky on the direct path: the by-band cloud optical depth is added where tau is stored (one pass less over the g-point array)
    tau_cld = be.cloud_optics_1scl(cloud_lut, atm.lwp, atm.iwp, atm.rel, atm.dei) if cloud_lut is not None else None
    fused = tau_cld if (direct and fuse_clouds) else None
    if direct:          # product default: interpolation recomputed inside the two consumers, no intermediate arrays
        if lite:        # tau and the Planck-lite outputs in one pass over the cells
            tau = be.empty((ngpt, nlay, ncol))
            fr = be.gas_optics_lw_fractions(kd, atm.p_lay, atm.t_lay, atm.t_lev, atm.t_sfc, _sfc_lay(atm), col_gas, tau, by_band=fused)
            src = dict(sfc_src=fr["sfc_src"], sfc_src_jac=fr["sfc_src_jac"], lay_src=None, lev_src=None)
            if keep:
                src["lay_src"], src["lev_src"] = be.planck_sources_from_fractions(kd, fr)
        else:
            tau = be.gas_optics_lw_direct(kd, atm.p_lay, atm.t_lay, col_gas, be.empty((ngpt, nlay, ncol)), by_band=fused)
            src = be.planck_source_direct(kd, atm.p_lay, atm.t_lay, atm.t_lev, atm.t_sfc, _sfc_lay(atm), col_gas)
    else:
        if hasattr(be, "compute_tau_absorption_set"):
            tau = be.compute_tau_absorption_set(kd, it, atm.p_lay, atm.t_lay, col_gas, be.empty((ngpt, nlay, ncol)))
        else:
            tau = be.zeros((ngpt, nlay, ncol))
            be.compute_tau_absorption(kd, it, atm.p_lay, atm.t_lay, col_gas, tau)
        src = be.compute_planck_source(kd, it, atm.t_lay, atm.t_lev, atm.t_sfc, _sfc_lay(atm))

    if tau_cld is not None and fused is None:
        be.inc_1scalar_by_1scalar_bybnd(tau, tau_cld, kd.band_lims_gpt)

    sfc_emis_gpt = be.expand_and_transpose(kd.band_lims_gpt, atm.emis_sfc, ngpt)
    gauss_Ds = be.asarray(GAUSS_DS)
    weights = be.asarray(np.ascontiguousarray(GAUSS_WTS[n_gauss_angles-1, :n_gauss_angles]))
    secants = be.lw_secants_array(ncol, ngpt, n_gauss_angles, MAX_GAUSS_PTS, gauss_Ds)

    if fr is not None:
        r = be.lw_solver_noscat_fractions(atm.top_at_1, kd, secants, weights, tau, fr, sfc_emis_gpt)
    else:
        r = be.lw_solver_noscat(atm.top_at_1, secants, weights, tau, src["lay_src"], src["lev_src"],
                                sfc_emis_gpt, src["sfc_src"], None, do_broadband=do_broadband)
    if do_broadband:
        flux_up, flux_dn = r["flux_up"], r["flux_dn"]
    else:
        flux_up = be.sum_broadband(r["flux_up"])
        flux_dn = be.sum_broadband(r["flux_dn"])
    flux_net = be.net_broadband_precalc(flux_dn, flux_up)
    out = dict(flux_up=flux_up, flux_dn=flux_dn, flux_net=flux_net)
    if keep:
        out.update(tau=tau, lay_src=src["lay_src"], lev_src=src["lev_src"], sfc_src=src["sfc_src"],
                   gpt_flux_up=r.get("flux_up"), gpt_flux_dn=r.get("flux_dn"), interp=it, col_gas=col_gas)
    return out


def solve_sw(be, kd, atm, col_dry=None, cloud_lut=None, delta_cloud=False, do_broadband=False, fused_gas=None, keep=False, direct=None,
             aerosol_lut=None, delta_aerosol=False, fuse_clouds=True):
    ncol, nlay, ngpt = atm.ncol, atm.nlay, kd.ngpt
    if fused_gas is None:
        fused_gas = hasattr(be, "gas_optics_sw_fused")
    direct = bool(fused_gas) and _use_direct(be, direct)
    col_dry, col_gas, it = gas_state(be, kd, atm, col_dry, interpolate=not direct)

    # clear sky: the asymmetry parameter of the gas optics is identically zero; the HIP entry points take "no g array"
    # natively (nothing written by the gas optics, nothing read by the solver)
    g_zero = bool(fused_gas and cloud_lut is None and aerosol_lut is None and getattr(be, "supports_null_g", False))
    cld = None
    if cloud_lut is not None:
        cld = be.cloud_optics_2str(cloud_lut, atm.lwp, atm.iwp, atm.rel, atm.dei)
        if delta_cloud:
            be.delta_scale_2str_k(*cld)
    fused = cld if (direct and fuse_clouds) else None          # added where the gas optics is stored
    if fused_gas:
        tau = be.empty((ngpt, nlay, ncol)); ssa = be.empty((ngpt, nlay, ncol))
        g = None if g_zero else be.empty((ngpt, nlay, ncol))
        if direct:
            be.gas_optics_sw_direct(kd, atm.p_lay, atm.t_lay, col_gas, col_dry, tau, ssa, g, by_band=fused)
        else:
            be.gas_optics_sw_fused(kd, it, atm.p_lay, atm.t_lay, col_gas, col_dry, tau, ssa, g)
    else:
        tau_abs = be.zeros((ngpt, nlay, ncol))
        be.compute_tau_absorption(kd, it, atm.p_lay, atm.t_lay, col_gas, tau_abs)
        tau_ray = be.compute_tau_rayleigh(kd, it, col_dry, col_gas)
        tau, ssa, g = be.combine_abs_and_rayleigh(tau_abs, tau_ray)

    toa_src = be.spread_col(ncol, kd.solar_source)
    be.scaling_to_subset(toa_src, atm.tsi_scaling)

    if cld is not None and fused is None:
        be.inc_2stream_by_2stream_bybnd(tau, ssa, g, *cld, kd.band_lims_gpt)

    if aerosol_lut is not None:
        # /root/reference/src_test/Radiation_solver.cu:794-809
        ta, wa, ga = be.aerosol_optics(aerosol_lut, [atm.aermr["aermr%02d" % i] for i in range(1, 12)], atm.rh, atm.p_lev)
        if delta_aerosol:
            be.delta_scale_2str_k(ta, wa, ga)
        be.inc_2stream_by_2stream_bybnd(tau, ssa, g, ta, wa, ga, kd.band_lims_gpt)

    alb_dir = be.expand_and_transpose(kd.band_lims_gpt, atm.sfc_alb_dir, ngpt)
    alb_dif = be.expand_and_transpose(kd.band_lims_gpt, atm.sfc_alb_dif, ngpt)

    r = be.sw_solver_2stream(atm.top_at_1, tau, ssa, g, atm.mu0, alb_dir, alb_dif, toa_src, None, do_broadband=do_broadband)
    if do_broadband:
        flux_up, flux_dn, flux_dir = r["flux_up"], r["flux_dn"], r["flux_dir"]
    else:
        flux_up = be.sum_broadband(r["flux_up"])
        flux_dn = be.sum_broadband(r["flux_dn"])
        flux_dir = be.sum_broadband(r["flux_dir"])
    flux_net = be.net_broadband_precalc(flux_dn, flux_up)
    out = dict(flux_up=flux_up, flux_dn=flux_dn, flux_dn_dir=flux_dir, flux_net=flux_net)
    if keep:
        out.update(tau=tau, ssa=ssa, g=be.zeros((ngpt, nlay, ncol)) if g is None else g, toa_src=toa_src, gpt_flux_up=r.get("flux_up"),
                   gpt_flux_dn=r.get("flux_dn"), gpt_flux_dir=r.get("flux_dir"))
    return out


class ResidentSolver:
    """One full clear-sky LW+SW solve per ``step()`` with every buffer allocated once (what a host model keeps
    resident between radiation calls, cf. the cached subsets in /root/reference/src_test/Radiation_solver.cu:450-466).
    HIP backend only; used by bench.py and the full-size tests. ``stage_events`` (torch.cuda.Event pairs on the
    launch stream) give per-stage device durations without synchronising inside the step."""

    STAGES = ("lw_gas_optics", "lw_planck", "lw_solver", "lw_reduce", "sw_gas_optics", "sw_solver", "sw_reduce")

    def __init__(self, be, kd_lw, kd_sw, atm, do_broadband=False, overlap=False, cloud_luts=None, sort_columns=None):
        import torch
        self.torch = torch
        # Column sorting. The windowed gas optics stages ONE box of LUT nodes per 256 neighbouring cells; columns
        # that differ much in pressure (RFMIP-like sites, --col-spread) do not fit one box and fall back to the gather kernels
        # (+40 % per step at +-35 % pressure spread). Columns are independent, so the step may process them in any order: sorted by
        # surface pressure, neighbours are alike again. The inputs are gathered into sorted order at the top of the step and the seven
        # broadband flux arrays scattered back at its end (0.5 ms per step at C4: ~30 gathers of (nlay, ncol) arrays). "auto" (default,
        # or RRX_SORT_COLUMNS): decided once from the initial atmosphere -- on when the surface pressure varies by more than 20 % inside
        # some 256-column block, about one cell of the LUT's pressure grid (ln p spacing 0.2): below that the boxes still fit and
        # sorting costs more than it brings (measured: +-5 % spread 15.1 ms unsorted, 15.6 sorted; +-35 %: 19.4 -> 16.0 ms).
        if sort_columns is None:
            sort_columns = os.environ.get("RRX_SORT_COLUMNS", "auto")
        if sort_columns == "auto":
            ps = atm.p_lev[-1 if atm.top_at_1 else 0]
            nb = (atm.ncol // 256) * 256
            if nb >= 256:
                blk = ps[:nb].reshape(-1, 256)
                sort_columns = bool((((blk.max(dim=1).values - blk.min(dim=1).values) / blk.mean(dim=1)) > 0.2).any().item())
            else:
                sort_columns = False
        self.sort_columns = bool(sort_columns) and str(sort_columns) != "0"
        # Padding (round 4): the step runs on a multiple of 16 columns -- the rows of the (col, lay, gpt) intermediates then start on
        # 128-B lines (16 385 columns cost 20 % more than 16 384 otherwise); the extra columns repeat the last one and are dropped
        # when the fluxes go back to the caller's order. One gather index serves both (built once, not per step, and
        # applied to an explicit list of per-column fields): refresh_column_order() rebuilds it when the host model's pressures
        # have changed enough to matter.
        pad = bool(int(os.environ.get("RRX_PAD_COLUMNS", "1"))) and atm.ncol > 16 and atm.ncol % 16 != 0
        self.npad = (16 - atm.ncol % 16) if pad else 0
        self.ncol_caller = atm.ncol
        self.perm = None
        self.atm = atm
        if self.sort_columns or self.npad:
            self.refresh_column_order()
        # overlap: LW and SW chains are independent, so they can run on two HIP streams and share the chip (the gather-
        # bound gas-optics kernels of one chain fill in next to the HBM-bound solver of the other)
        self.overlap = overlap
        self.streams = [torch.cuda.Stream(device=be.device), torch.cuda.Stream(device=be.device)] if overlap else None
        self.be, self.kd_lw, self.kd_sw, self.atm = be, kd_lw, kd_sw, atm
        self.do_broadband = do_broadband
        # all-sky (BASELINE C5): (lw_lut, sw_lut) of cloud optics; clouds are added by band after the gas optics, delta-scaled in SW
        self.cloud_luts = cloud_luts
        self.fuse_clouds = bool(int(os.environ.get("RRX_FUSE_CLOUDS", "1")))
        self.g_zero = cloud_luts is None and bool(int(os.environ.get("RRX_G_ZERO", "1")))     # clear sky: g == 0 is neither written nor read
        self.direct = bool(int(os.environ.get("RRX_DIRECT", "1")))     # interpolation state recomputed inside its consumers
        # broadband mode: Planck fractions + band Planck functions, sources formed inside the LW solver ("Planck-lite" chain)
        self.lite = self.direct and do_broadband and bool(int(os.environ.get("RRX_LITE", "1")))
        ncol, nlay = atm.ncol + self.npad, atm.nlay              # columns of a step (padded)
        ng_l, ng_s = kd_lw.ngpt, kd_sw.ngpt
        e = be.empty
        self.col_dry = e((nlay, ncol))
        if self.lite:
            self.lw = dict(tau=e((ng_l, nlay, ncol)), pfrac=e((ng_l, nlay, ncol)), blay=e((kd_lw.nbnd, nlay, ncol)),
                           blev=e((kd_lw.nbnd, nlay+1, ncol)), sfc_src=e((ng_l, ncol)), sfc_src_jac=e((ng_l, ncol)))
        else:
            self.lw = dict(tau=e((ng_l, nlay, ncol)), lay_src=e((ng_l, nlay, ncol)), lev_src=e((ng_l, nlay+1, ncol)),
                           sfc_src=e((ng_l, ncol)), sfc_src_jac=e((ng_l, ncol)))
        self.sw = dict(tau=e((ng_s, nlay, ncol)), ssa=e((ng_s, nlay, ncol)), g=e((ng_s, nlay, ncol)))
        if not do_broadband:
            self.lw.update(gpt_up=e((ng_l, nlay+1, ncol)), gpt_dn=e((ng_l, nlay+1, ncol)))
            self.sw.update(gpt_up=e((ng_s, nlay+1, ncol)), gpt_dn=e((ng_s, nlay+1, ncol)), gpt_dir=e((ng_s, nlay+1, ncol)))
        # packed broadband outputs: LW up/dn/net + SW up/dn/dir/net  (7, nlev, ncol) -> one all-gather
        self.fluxes = e((7, nlay+1, atm.ncol))
        self.fluxes_sorted = e((7, nlay+1, ncol)) if self.perm is not None else None
        self.weights = be.asarray(np.ascontiguousarray(GAUSS_WTS[0, :1]))
        self.gauss_Ds = be.asarray(GAUSS_DS)
        # secants and the band -> g-point expansion of emissivity / albedos: buffers allocated once, the launches themselves are
        # part of every step (rte_lw / rte_sw of the reference run them per solve: src_cuda/Rte_lw.cu:70-110, Rte_sw.cu:57-100)
        self.secants = e((1, ng_l, ncol))
        self.sfc_emis_gpt = e((ng_l, ncol)); self.alb_dir = e((ng_s, ncol)); self.alb_dif = e((ng_s, ncol))
        self.events = None
        self.col_dry2 = None

    def enable_stage_events(self, nsteps):
        ev = self.torch.cuda.Event
        self.events = [{s: (ev(enable_timing=True), ev(enable_timing=True)) for s in self.STAGES} for _ in range(nsteps)]
        self._istep = 0

    def stage_ms(self):
        out = {s: [] for s in self.STAGES}
        for rec in self.events[:self._istep]:
            for s, (a, b) in rec.items():
                out[s].append(a.elapsed_time(b))
        return {s: float(np.mean(v)) for s, v in out.items() if v}

    # per-column fields of an Atmosphere and the axis their column index sits on (everything else is shared by all columns)
    _COLUMN_FIELDS = {"p_lay": -1, "p_lev": -1, "t_lay": -1, "t_lev": -1, "t_sfc": 0, "mu0": 0, "tsi_scaling": 0,
                      "emis_sfc": 0, "sfc_alb_dir": 0, "sfc_alb_dif": 0, "lwp": -1, "iwp": -1, "rel": -1, "dei": -1, "rh": -1}

    def refresh_column_order(self):
        """(Re)build the gather index of a step: ascending surface pressure when sorting, the caller's order otherwise, padded with
        repeats of its last entry. Call it again when the pressures have changed enough to matter; the index is reused otherwise."""
        torch = self.torch
        a = self.atm
        if self.sort_columns:
            perm = torch.argsort(a.p_lev[-1 if a.top_at_1 else 0])
        else:
            perm = torch.arange(a.ncol, device=a.p_lev.device)
        if self.npad:
            perm = torch.cat([perm, perm[-1:].expand(self.npad)])
        self.perm = perm.contiguous()

    def _gathered_atmosphere(self):
        """The atmosphere with its columns in the order (and count) of self.perm."""
        from .synthetic import Atmosphere
        a, perm = self.atm, self.perm
        out = {}
        for k, v in a.__dict__.items():
            if k == "ncol":
                out[k] = int(perm.numel())
            elif k in self._COLUMN_FIELDS and v is not None:
                out[k] = v.index_select(v.dim() - 1 if self._COLUMN_FIELDS[k] < 0 else 0, perm)
            elif isinstance(v, dict):      # vmr / aermr: (nlay, ncol) fields are per column, profiles (nlay,) and scalars are shared
                out[k] = {n: (t.index_select(1, perm) if (hasattr(t, "dim") and t.dim() == 2) else t) for n, t in v.items()}
            else:
                out[k] = v
        return Atmosphere(**out)

    def step(self):
        be, atm = self.be, self.atm
        perm = self.perm
        if perm is not None:
            atm = self._gathered_atmosphere()
        if self.overlap and self.col_dry2 is None:
            self.col_dry2 = be.empty(tuple(self.col_dry.shape))
        rec = None
        if self.events is not None and self._istep < len(self.events):
            rec = self.events[self._istep]
            self._istep += 1

        def mark(stage, end=False):
            if rec is not None:
                rec[stage][1 if end else 0].record(self.torch.cuda.current_stream(be.device))   # the chain's stream when overlapping

        ncol, nlay = atm.ncol, atm.nlay
        F = self.fluxes if perm is None else self.fluxes_sorted
        main = self.torch.cuda.current_stream(be.device)
        for ichain, (kind, kd, buf) in enumerate((("lw", self.kd_lw, self.lw), ("sw", self.kd_sw, self.sw))):
            if self.overlap:
                self.streams[ichain].wait_stream(main)
                ctx = self.torch.cuda.stream(self.streams[ichain])
                ctx.__enter__()
            mark(kind + "_gas_optics")
            col_dry = self.col_dry2 if (self.overlap and ichain == 1) else self.col_dry
            be._c("get_col_dry", ncol, nlay, atm.vmr["h2o"], atm.p_lev, col_dry)
            col_gas = be.fill_gases(kd, atm.vmr, col_dry)
            it = None if self.direct else be.interpolation(kd, atm.p_lay, atm.t_lay, col_gas)
            # all-sky: the by-band cloud properties are computed first and added where the gas optics is stored (fused); with
            # RRX_FUSE_CLOUDS=0 by the reference's separate increment kernels afterwards
            fuse = self.cloud_luts is not None and self.direct and self.fuse_clouds
            if kind == "lw":
                tc = None
                if self.cloud_luts is not None:     # /root/reference/src_test/Radiation_solver.cu:497-512
                    tc = be.cloud_optics_1scl(self.cloud_luts[0], atm.lwp, atm.iwp, atm.rel, atm.dei)
                if self.lite:
                    be.gas_optics_lw_fractions(kd, atm.p_lay, atm.t_lay, atm.t_lev, atm.t_sfc, _sfc_lay(atm), col_gas, buf["tau"], out=buf,
                                               by_band=tc if fuse else None)
                elif self.direct:
                    be.gas_optics_lw_direct(kd, atm.p_lay, atm.t_lay, col_gas, buf["tau"], by_band=tc if fuse else None)
                else:
                    be.compute_tau_absorption_set(kd, it, atm.p_lay, atm.t_lay, col_gas, buf["tau"])
                if tc is not None and not fuse:
                    be.inc_1scalar_by_1scalar_bybnd(buf["tau"], tc, kd.band_lims_gpt)
                mark("lw_gas_optics", True)
                mark("lw_planck")
                srcs = None if self.lite else dict(sfc_src=buf["sfc_src"], lay_src=buf["lay_src"], lev_src=buf["lev_src"], sfc_src_jac=buf["sfc_src_jac"])
                if self.lite:
                    pass                                   # the fractions came with the optical depths
                elif self.direct:
                    be.planck_source_direct(kd, atm.p_lay, atm.t_lay, atm.t_lev, atm.t_sfc, _sfc_lay(atm), col_gas, out=srcs)
                else:
                    be.compute_planck_source(kd, it, atm.t_lay, atm.t_lev, atm.t_sfc, _sfc_lay(atm), out=srcs)
                mark("lw_planck", True)
                mark("lw_solver")
                be._c("lw_secants_array", ncol, kd.ngpt, 1, MAX_GAUSS_PTS, self.gauss_Ds, self.secants)
                be._c("expand_and_transpose", ncol, kd.nbnd, kd.band_lims_gpt, atm.emis_sfc, self.sfc_emis_gpt)
                if self.lite:
                    be.lw_solver_noscat_fractions(atm.top_at_1, kd, self.secants, self.weights, buf["tau"], buf, self.sfc_emis_gpt,
                                                  flux_up=F[0], flux_dn=F[1])
                elif self.do_broadband:
                    be._c("lw_solver_noscat", ncol, nlay, kd.ngpt, BoolArg(atm.top_at_1), 1, self.secants, self.weights,
                          buf["tau"], buf["lay_src"], buf["lev_src"], self.sfc_emis_gpt, buf["sfc_src"], None, None, None,
                          BoolArg(True), F[0], F[1], BoolArg(False), None, None)
                else:
                    be.lw_solver_noscat_into(atm.top_at_1, self.secants, self.weights, buf["tau"], buf["lay_src"],
                                             buf["lev_src"], self.sfc_emis_gpt, buf["sfc_src"], buf["gpt_up"], buf["gpt_dn"])
                mark("lw_solver", True)
                mark("lw_reduce")
                if not self.do_broadband:
                    be.sum_broadband(buf["gpt_up"], out=F[0]); be.sum_broadband(buf["gpt_dn"], out=F[1])
                be.net_broadband_precalc(F[1], F[0], out=F[2])
                mark("lw_reduce", True)
            else:
                # clear sky: the asymmetry parameter is identically zero; the fused broadband solver takes "no g" natively
                gbuf = None if (self.g_zero and self.do_broadband) else buf["g"]
                cld = None
                if self.cloud_luts is not None:     # Radiation_solver.cu:773-792
                    cld = be.cloud_optics_2str(self.cloud_luts[1], atm.lwp, atm.iwp, atm.rel, atm.dei, delta_scale=True)
                if self.direct:
                    be.gas_optics_sw_direct(kd, atm.p_lay, atm.t_lay, col_gas, col_dry, buf["tau"], buf["ssa"], gbuf, by_band=cld if fuse else None)
                else:
                    be.gas_optics_sw_fused(kd, it, atm.p_lay, atm.t_lay, col_gas, col_dry, buf["tau"], buf["ssa"], gbuf)
                toa = be.toa_source(ncol, kd.solar_source, atm.tsi_scaling)     # spread_col + scaling_to_subset, one launch
                if cld is not None and not fuse:
                    be.inc_2stream_by_2stream_bybnd(buf["tau"], buf["ssa"], gbuf, *cld, kd.band_lims_gpt)
                mark("sw_gas_optics", True)
                mark("sw_solver")
                be._c("expand_and_transpose", ncol, kd.nbnd, kd.band_lims_gpt, atm.sfc_alb_dir, self.alb_dir)
                be._c("expand_and_transpose", ncol, kd.nbnd, kd.band_lims_gpt, atm.sfc_alb_dif, self.alb_dif)
                if self.do_broadband:
                    be._c("sw_solver_2stream", ncol, nlay, kd.ngpt, BoolArg(atm.top_at_1), buf["tau"], buf["ssa"], gbuf, atm.mu0,
                          self.alb_dir, self.alb_dif, toa, None, None, None, BoolArg(False), None,
                          BoolArg(True), F[3], F[4], F[5])
                else:
                    be.sw_solver_2stream_into(atm.top_at_1, buf["tau"], buf["ssa"], gbuf, atm.mu0, self.alb_dir, self.alb_dif,
                                              toa, buf["gpt_up"], buf["gpt_dn"], buf["gpt_dir"])
                mark("sw_solver", True)
                mark("sw_reduce")
                if not self.do_broadband:
                    be.sum_broadband(buf["gpt_up"], out=F[3]); be.sum_broadband(buf["gpt_dn"], out=F[4]); be.sum_broadband(buf["gpt_dir"], out=F[5])
                be.net_broadband_precalc(F[4], F[3], out=F[6])
                mark("sw_reduce", True)
            if self.overlap:
                ctx.__exit__(None, None, None)
        if self.overlap:
            main.wait_stream(self.streams[0]); main.wait_stream(self.streams[1])
        if perm is not None:                              # back to the caller's column order (padding columns dropped)
            n = self.ncol_caller
            self.fluxes.index_copy_(2, perm[:n], F[:, :, :n])
            F = self.fluxes
        return F
