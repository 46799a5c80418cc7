"""Python view of the C ABI of librrx_hip.so (include/rrx_hip.h), operating on torch CUDA tensors.

This is plumbing for tests and bench.py: torch supplies device memory, streams and torch.distributed; every
number is produced by the hand-written HIP kernels behind the C ABI. There is NO CPU fallback: importing this
module without the built library, or calling it without a GPU, raises.

Array convention (see synthetic.py): tensors are C-contiguous with reversed dimensions, so their memory is the
reference's column-major layout, e.g. tau(ncol,nlay,ngpt) <-> tensor shape (ngpt, nlay, ncol).
"""
import ctypes
import os
import numpy as np
import torch

from ._ffi import Lib, BoolArg, PtrArg

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librrx_hip.so")


def load_library():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The product path has no CPU fallback.")
    return ctypes.CDLL(LIB_PATH)


class HipKernels:
    """Backend object with the launcher-level interface used by pipeline.py (same method names as the
    test-only CPU backends in oracle/oracle_py.py)."""
    name = "hip"

    def __init__(self, dtype=np.float64, device="cuda:0", stream=None):
        if not torch.cuda.is_available():
            raise RuntimeError("HipKernels needs a GPU (no CPU fallback on the product path)")
        self.np_dtype = np.dtype(dtype)
        self.sfx = "_f64" if self.np_dtype == np.float64 else "_f32"
        self.tdtype = torch.float64 if self.np_dtype == np.float64 else torch.float32
        self.device = torch.device(device)
        cdll = load_library()
        cdll.rrx_last_error.restype = ctypes.c_char_p
        self.lib = Lib(LIB_PATH, self.np_dtype, by_ref=False, returns_status=True,
                       error_fn=lambda: cdll.rrx_last_error().decode())
        self.stream = stream          # torch.cuda.Stream or None (= torch current stream)

    # ---- memory helpers -------------------------------------------------------------------------
    def _st(self):
        s = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        return PtrArg(s.cuda_stream)

    def asarray(self, a):
        if isinstance(a, torch.Tensor):
            return a
        a = np.ascontiguousarray(a)
        if a.dtype.kind == "f":
            a = a.astype(self.np_dtype)
        return torch.from_numpy(a).to(self.device)

    def to_numpy(self, t):
        return t.detach().cpu().numpy()

    def empty(self, shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.tdtype, device=self.device)

    def zeros(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.tdtype, device=self.device)

    def int_empty(self, shape):
        return torch.empty(shape, dtype=torch.int32, device=self.device)

    def bool_empty(self, shape):
        return torch.empty(shape, dtype=torch.int8, device=self.device)

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    def upload_kdist(self, kd):
        from .synthetic import KDist
        out = {}
        for k, v in kd.__dict__.items():
            out[k] = self.asarray(v) if isinstance(v, np.ndarray) else v
        return KDist(**out)

    def _c(self, name, *args):
        return self.lib.call("rrx_" + name + self.sfx, *args, self._st())

    # ---- solvers ----------------------------------------------------------------------------------
    def lw_secants_array(self, ncol, ngpt, n_quad, max_pts, gauss_Ds):
        sec = self.empty((n_quad, ngpt, ncol))
        self._c("lw_secants_array", ncol, ngpt, n_quad, max_pts, gauss_Ds, sec)
        return sec

    def lw_solver_noscat(self, top_at_1, secants, weights, tau, lay_source, lev_source, sfc_emis, sfc_src,
                         inc_flux=None, do_broadband=False, do_jacobians=False, sfc_src_jac=None):
        ngpt, nlay, ncol = tau.shape
        nmus = weights.shape[0]
        out = {}
        flux_up = flux_dn = up_loc = dn_loc = jac = None
        if do_broadband:
            up_loc = self.empty((nlay+1, ncol)); dn_loc = self.empty((nlay+1, ncol))
            out.update(flux_up=up_loc, flux_dn=dn_loc)
        else:
            flux_up = self.empty((ngpt, nlay+1, ncol)); flux_dn = self.empty((ngpt, nlay+1, ncol))
            out.update(flux_up=flux_up, flux_dn=flux_dn)
        if do_jacobians:
            jac = self.empty((ngpt, nlay+1, ncol))
            out["flux_up_jac"] = jac
        self._c("lw_solver_noscat", ncol, nlay, ngpt, BoolArg(top_at_1), nmus, secants, weights,
                tau, lay_source, lev_source, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn,
                BoolArg(do_broadband), up_loc, dn_loc, BoolArg(do_jacobians), sfc_src_jac, jac)
        return out

    def lw_solver_noscat_into(self, top_at_1, secants, weights, tau, lay_source, lev_source, sfc_emis, sfc_src,
                              flux_up, flux_dn):
        """Allocation-free form used inside timed regions."""
        ngpt, nlay, ncol = tau.shape
        self._c("lw_solver_noscat", ncol, nlay, ngpt, BoolArg(top_at_1), weights.shape[0], secants, weights,
                tau, lay_source, lev_source, sfc_emis, sfc_src, None, flux_up, flux_dn,
                BoolArg(False), None, None, BoolArg(False), None, None)

    def sw_solver_2stream(self, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif, inc_flux_dir,
                          inc_flux_dif=None, do_broadband=False):
        ngpt, nlay, ncol = tau.shape
        if mu0.dim() == 2:          # (nlay, ncol) CPU-style: the GPU boundary takes mu0(ncol)
            mu0 = mu0[0].contiguous()
        fu = fd = fr = ul = dl = rl = None
        if do_broadband:
            ul = self.empty((nlay+1, ncol)); dl = self.empty((nlay+1, ncol)); rl = self.empty((nlay+1, ncol))
            out = dict(flux_up=ul, flux_dn=dl, flux_dir=rl)
        else:
            fu = self.empty((ngpt, nlay+1, ncol)); fd = self.empty((ngpt, nlay+1, ncol)); fr = self.empty((ngpt, nlay+1, ncol))
            out = dict(flux_up=fu, flux_dn=fd, flux_dir=fr)
        self._c("sw_solver_2stream", ncol, nlay, ngpt, BoolArg(top_at_1), tau, ssa, g, mu0,
                sfc_alb_dir, sfc_alb_dif, inc_flux_dir, fu, fd, fr,
                BoolArg(inc_flux_dif is not None), inc_flux_dif, BoolArg(do_broadband), ul, dl, rl)
        return out

    def sw_solver_2stream_into(self, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif, inc_flux_dir,
                               flux_up, flux_dn, flux_dir):
        ngpt, nlay, ncol = tau.shape
        self._c("sw_solver_2stream", ncol, nlay, ngpt, BoolArg(top_at_1), tau, ssa, g, mu0,
                sfc_alb_dir, sfc_alb_dif, inc_flux_dir, flux_up, flux_dn, flux_dir,
                BoolArg(False), None, BoolArg(False), None, None, None)

    # ---- gas optics -------------------------------------------------------------------------------
    def interpolation(self, kd, play, tlay, col_gas):
        nlay, ncol = play.shape
        r = dict(
            jtemp=self.int_empty((nlay, ncol)), jpress=self.int_empty((nlay, ncol)),
            tropo=self.bool_empty((nlay, ncol)),
            jeta=self.int_empty((kd.nflav, nlay, ncol, 2)),
            col_mix=self.empty((kd.nflav, nlay, ncol, 2)),
            fminor=self.empty((kd.nflav, nlay, ncol, 2, 2)),
            fmajor=self.empty((kd.nflav, nlay, ncol, 2, 2, 2)))
        self._c("interpolation", ncol, nlay, kd.ngas, kd.nflav, kd.neta, kd.npres, kd.ntemp,
                kd.flavor, kd.press_ref_log, kd.temp_ref,
                float(kd.press_ref_log_delta), float(kd.temp_ref_min), float(kd.temp_ref_delta), float(kd.press_ref_trop_log),
                kd.vmr_ref, play, tlay, col_gas,
                r["jtemp"], r["fmajor"], r["fminor"], r["col_mix"], r["tropo"], r["jeta"], r["jpress"])
        return r

    def _absorption_args(self, kd, it, play, tlay, col_gas):
        nlay, ncol = play.shape
        return (ncol, nlay, kd.nbnd, kd.ngpt, kd.ngas, kd.nflav, kd.neta, kd.npres, kd.ntemp,
                kd.minor_limits_gpt_lower.shape[0], kd.kminor_lower.shape[0],
                kd.minor_limits_gpt_upper.shape[0], kd.kminor_upper.shape[0], kd.idx_h2o,
                kd.gpoint_flavor, kd.band_lims_gpt, kd.kmajor, kd.kminor_lower, kd.kminor_upper,
                kd.minor_limits_gpt_lower, kd.minor_limits_gpt_upper,
                kd.minor_scales_with_density_lower, kd.minor_scales_with_density_upper,
                kd.scale_by_complement_lower, kd.scale_by_complement_upper,
                kd.idx_minor_lower, kd.idx_minor_upper, kd.idx_minor_scaling_lower, kd.idx_minor_scaling_upper,
                kd.kminor_start_lower, kd.kminor_start_upper,
                it["tropo"], it["col_mix"], it["fmajor"], it["fminor"], play, tlay, col_gas)

    def compute_tau_absorption(self, kd, it, play, tlay, col_gas, tau):
        self._c("compute_tau_absorption", *self._absorption_args(kd, it, play, tlay, col_gas),
                it["jeta"], it["jtemp"], it["jpress"], tau)
        return tau

    def compute_tau_absorption_set(self, kd, it, play, tlay, col_gas, tau):
        """tau = major + minor (no zero fill, no read-back); same arithmetic as compute_tau_absorption on a zeroed tau"""
        self._c("compute_tau_absorption_set", *self._absorption_args(kd, it, play, tlay, col_gas),
                it["jeta"], it["jtemp"], it["jpress"], tau)
        return tau

    def gas_optics_sw_fused(self, kd, it, play, tlay, col_gas, col_dry, tau, ssa, g):
        self._c("gas_optics_sw_fused", *self._absorption_args(kd, it, play, tlay, col_gas), col_dry,
                it["jeta"], it["jtemp"], it["jpress"], kd.krayl, tau, ssa, g)

    # "direct" forms: interpolation state computed inside the consumer (no intermediate arrays)
    def _direct_args(self, kd):
        return (kd.flavor, kd.press_ref_log, kd.temp_ref, float(kd.press_ref_log_delta), float(kd.temp_ref_min),
                float(kd.temp_ref_delta), float(kd.press_ref_trop_log), kd.vmr_ref)

    def _minor_args(self, kd, play):
        nlay, ncol = play.shape
        return (ncol, nlay, kd.nbnd, kd.ngpt, kd.ngas, kd.nflav, kd.neta, kd.npres, kd.ntemp,
                kd.minor_limits_gpt_lower.shape[0], kd.kminor_lower.shape[0],
                kd.minor_limits_gpt_upper.shape[0], kd.kminor_upper.shape[0], kd.idx_h2o,
                kd.gpoint_flavor, kd.band_lims_gpt, kd.kmajor, kd.kminor_lower, kd.kminor_upper,
                kd.minor_limits_gpt_lower, kd.minor_limits_gpt_upper,
                kd.minor_scales_with_density_lower, kd.minor_scales_with_density_upper,
                kd.scale_by_complement_lower, kd.scale_by_complement_upper,
                kd.idx_minor_lower, kd.idx_minor_upper, kd.idx_minor_scaling_lower, kd.idx_minor_scaling_upper,
                kd.kminor_start_lower, kd.kminor_start_upper)

    def gas_optics_lw_direct(self, kd, play, tlay, col_gas, tau, by_band=None):
        """by_band: optical depth per band (nbnd, nlay, ncol) -- clouds -- added where tau is stored (the _allsky entry)"""
        if by_band is None:
            self._c("gas_optics_lw_direct", *self._minor_args(kd, play), *self._direct_args(kd), play, tlay, col_gas, tau)
        else:
            self._c("gas_optics_lw_direct_allsky", *self._minor_args(kd, play), *self._direct_args(kd), play, tlay, col_gas, tau, by_band)
        return tau

    def gas_optics_sw_direct(self, kd, play, tlay, col_gas, col_dry, tau, ssa, g, by_band=None):
        """by_band: (tau, ssa, g) per band (nbnd, nlay, ncol) -- clouds, aerosols -- combined with the gas optics where it is stored"""
        if by_band is None:
            self._c("gas_optics_sw_direct", *self._minor_args(kd, play), *self._direct_args(kd), play, tlay, col_gas, col_dry,
                    kd.krayl, tau, ssa, g)
        else:
            self._c("gas_optics_sw_direct_allsky", *self._minor_args(kd, play), *self._direct_args(kd), play, tlay, col_gas, col_dry,
                    kd.krayl, tau, ssa, g, *by_band)

    def planck_source_direct(self, kd, play, tlay, tlev, tsfc, sfc_lay, col_gas, out=None):
        nlay, ncol = tlay.shape
        if out is None:
            out = dict(sfc_src=self.empty((kd.ngpt, ncol)), lay_src=self.empty((kd.ngpt, nlay, ncol)),
                       lev_src=self.empty((kd.ngpt, nlay+1, ncol)), sfc_src_jac=self.empty((kd.ngpt, ncol)))
        self._c("planck_source_direct", ncol, nlay, kd.nbnd, kd.ngpt, kd.ngas, kd.nflav, kd.neta, kd.npres, kd.ntemp, kd.nPlanckTemp,
                play, tlay, tlev, tsfc, sfc_lay, col_gas, *self._direct_args(kd),
                kd.gpoint_bands, kd.band_lims_gpt, kd.planck_frac, float(kd.totplnk_delta), kd.totplnk, kd.gpoint_flavor,
                out["sfc_src"], out["lay_src"], out["lev_src"], out["sfc_src_jac"])
        return out

    # "Planck-lite" LW chain: fractions + band Planck functions, sources formed inside the broadband solver
    def planck_fractions(self, kd, play, tlay, tlev, tsfc, sfc_lay, col_gas, out=None):
        nlay, ncol = tlay.shape
        if out is None:
            out = dict(pfrac=self.empty((kd.ngpt, nlay, ncol)), blay=self.empty((kd.nbnd, nlay, ncol)),
                       blev=self.empty((kd.nbnd, nlay+1, ncol)), sfc_src=self.empty((kd.ngpt, ncol)), sfc_src_jac=self.empty((kd.ngpt, ncol)))
        self._c("planck_fractions", ncol, nlay, kd.nbnd, kd.ngpt, kd.ngas, kd.nflav, kd.neta, kd.npres, kd.ntemp, kd.nPlanckTemp,
                play, tlay, tlev, tsfc, sfc_lay, col_gas, *self._direct_args(kd),
                kd.gpoint_bands, kd.band_lims_gpt, kd.planck_frac, float(kd.totplnk_delta), kd.totplnk, kd.gpoint_flavor,
                out["pfrac"], out["blay"], out["blev"], out["sfc_src"], out["sfc_src_jac"])
        return out

    def gas_optics_lw_fractions(self, kd, play, tlay, tlev, tsfc, sfc_lay, col_gas, tau, out=None, by_band=None):
        """tau and the Planck-lite outputs in one pass (rrx_gas_optics_lw_fractions); by_band as in gas_optics_lw_direct"""
        nlay, ncol = tlay.shape
        if out is None:
            out = dict(pfrac=self.empty((kd.ngpt, nlay, ncol)), blay=self.empty((kd.nbnd, nlay, ncol)),
                       blev=self.empty((kd.nbnd, nlay+1, ncol)), sfc_src=self.empty((kd.ngpt, ncol)), sfc_src_jac=self.empty((kd.ngpt, ncol)))
        a = self._minor_args(kd, play)
        extra = () if by_band is None else (by_band,)
        self._c("gas_optics_lw_fractions" + ("" if by_band is None else "_allsky"), *a[:9], kd.nPlanckTemp, *a[9:16], kd.gpoint_bands, *a[16:],
                *self._direct_args(kd), play, tlay, tlev, tsfc, sfc_lay, col_gas, kd.planck_frac, float(kd.totplnk_delta), kd.totplnk,
                tau, out["pfrac"], out["blay"], out["blev"], out["sfc_src"], out["sfc_src_jac"], *extra)
        return out

    def planck_sources_from_fractions(self, kd, fr, lay_src=None, lev_src=None):
        ngpt, nlay, ncol = fr["pfrac"].shape
        lay_src = self.empty((ngpt, nlay, ncol)) if lay_src is None else lay_src
        lev_src = self.empty((ngpt, nlay+1, ncol)) if lev_src is None else lev_src
        self._c("planck_sources_from_fractions", ncol, nlay, ngpt, kd.gpoint_bands, fr["pfrac"], fr["blay"], fr["blev"], lay_src, lev_src)
        return lay_src, lev_src

    def lw_solver_noscat_fractions(self, top_at_1, kd, secants, weights, tau, fr, sfc_emis, inc_flux=None, flux_up=None, flux_dn=None):
        ngpt, nlay, ncol = tau.shape
        flux_up = self.empty((nlay+1, ncol)) if flux_up is None else flux_up
        flux_dn = self.empty((nlay+1, ncol)) if flux_dn is None else flux_dn
        self._c("lw_solver_noscat_fractions", ncol, nlay, ngpt, BoolArg(top_at_1), secants, weights, tau,
                fr["pfrac"], fr["blay"], fr["blev"], kd.gpoint_bands, sfc_emis, fr["sfc_src"], inc_flux, flux_up, flux_dn)
        return dict(flux_up=flux_up, flux_dn=flux_dn)

    def compute_tau_rayleigh(self, kd, it, col_dry, col_gas):
        nlay, ncol = col_dry.shape
        tr = self.empty((kd.ngpt, nlay, ncol))
        self._c("compute_tau_rayleigh", ncol, nlay, kd.nbnd, kd.ngpt, kd.ngas, kd.nflav, kd.neta, kd.npres, kd.ntemp,
                kd.gpoint_flavor, kd.band_lims_gpt, kd.krayl, kd.idx_h2o, col_dry, col_gas,
                it["fminor"], it["jeta"], it["tropo"], it["jtemp"], tr)
        return tr

    def combine_abs_and_rayleigh(self, tau_abs, tau_ray):
        ngpt, nlay, ncol = tau_abs.shape
        tau = self.empty(tau_abs.shape); ssa = self.empty(tau_abs.shape); g = self.empty(tau_abs.shape)
        self._c("combine_abs_and_rayleigh", ncol, nlay, ngpt, tau_abs, tau_ray, tau, ssa, g)
        return tau, ssa, g

    def compute_planck_source(self, kd, it, tlay, tlev, tsfc, sfc_lay, out=None):
        nlay, ncol = tlay.shape
        if out is None:
            out = dict(sfc_src=self.empty((kd.ngpt, ncol)), lay_src=self.empty((kd.ngpt, nlay, ncol)),
                       lev_src=self.empty((kd.ngpt, nlay+1, ncol)), sfc_src_jac=self.empty((kd.ngpt, ncol)))
        self._c("compute_planck_source", ncol, nlay, kd.nbnd, kd.ngpt, kd.nflav, kd.neta, kd.npres, kd.ntemp, kd.nPlanckTemp,
                tlay, tlev, tsfc, sfc_lay, it["fmajor"], it["jeta"], it["tropo"], it["jtemp"], it["jpress"],
                kd.gpoint_bands, kd.band_lims_gpt, kd.planck_frac, float(kd.temp_ref_min), float(kd.totplnk_delta),
                kd.totplnk, kd.gpoint_flavor, out["sfc_src"], out["lay_src"], out["lev_src"], out["sfc_src_jac"])
        return out

    # ---- stand-alone boundary conditions and transposes (launcher parity with the reference's namespaces) ----
    def apply_BC(self, nlay, top_at_1, flux_dn, inc_flux=None, factor=None):
        ngpt, nlev, ncol = flux_dn.shape
        if inc_flux is None:
            self._c("apply_BC_0", ncol, nlay, ngpt, BoolArg(top_at_1), flux_dn)
        elif factor is None:
            self._c("apply_BC_gpt", ncol, nlay, ngpt, BoolArg(top_at_1), inc_flux, flux_dn)
        else:
            self._c("apply_BC_factor", ncol, nlay, ngpt, BoolArg(top_at_1), inc_flux, factor, flux_dn)
        return flux_dn

    def reorder123x321(self, arr_in):
        ni, nj, nk = arr_in.shape
        out = self.empty((nk, nj, ni))
        self._c("reorder123x321", ni, nj, nk, arr_in, out)
        return out

    def reorder12x21(self, arr_in):
        ni, nj = arr_in.shape
        out = self.empty((nj, ni))
        self._c("reorder12x21", ni, nj, arr_in, out)
        return out

    # ---- optical props / fluxes -----------------------------------------------------------------------
    def increment_1scalar_by_1scalar(self, tau_inout, tau_in):
        ngpt, nlay, ncol = tau_inout.shape
        self._c("increment_1scalar_by_1scalar", ncol, nlay, ngpt, tau_inout, tau_in)

    def increment_2stream_by_2stream(self, t1, w1, g1, t2, w2, g2):
        ngpt, nlay, ncol = t1.shape
        self._c("increment_2stream_by_2stream", ncol, nlay, ngpt, t1, w1, g1, t2, w2, g2)

    def inc_1scalar_by_1scalar_bybnd(self, tau_inout, tau_in, band_lims):
        ngpt, nlay, ncol = tau_inout.shape
        self._c("inc_1scalar_by_1scalar_bybnd", ncol, nlay, ngpt, tau_inout, tau_in, band_lims.shape[0], band_lims)

    def inc_2stream_by_2stream_bybnd(self, t1, w1, g1, t2, w2, g2, band_lims):
        ngpt, nlay, ncol = t1.shape
        self._c("inc_2stream_by_2stream_bybnd", ncol, nlay, ngpt, t1, w1, g1, t2, w2, g2, band_lims.shape[0], band_lims)

    def delta_scale_2str_k(self, tau, ssa, g):
        ngpt, nlay, ncol = tau.shape
        self._c("delta_scale_2str_k", ncol, nlay, ngpt, tau, ssa, g)

    def sum_broadband(self, gpt_flux, out=None):
        ngpt, nlev, ncol = gpt_flux.shape
        out = self.empty((nlev, ncol)) if out is None else out
        self._c("sum_broadband", ncol, nlev, ngpt, gpt_flux, out)
        return out

    def net_broadband_precalc(self, flux_dn, flux_up, out=None):
        nlev, ncol = flux_dn.shape
        out = self.empty((nlev, ncol)) if out is None else out
        self._c("net_broadband_precalc", ncol, nlev, flux_dn, flux_up, out)
        return out

    def heating_rate(self, flux_net, plev, g_over_cp=9.80665/1004.64):
        nlev, ncol = flux_net.shape
        out = self.empty((nlev-1, ncol))
        self._c("heating_rate", ncol, nlev-1, float(g_over_cp), flux_net, plev, out)
        return out

    def sum_byband(self, gpt_flux, band_lims):
        ngpt, nlev, ncol = gpt_flux.shape
        nbnd = band_lims.shape[0]
        out = self.empty((nbnd, nlev, ncol))
        self._c("sum_byband", ncol, nlev, ngpt, nbnd, band_lims, gpt_flux, out)
        return out

    def net_byband_full(self, gpt_dn, gpt_up, band_lims):
        ngpt, nlev, ncol = gpt_dn.shape
        nbnd = band_lims.shape[0]
        out = self.empty((nbnd, nlev, ncol))
        self._c("net_byband_full", ncol, nlev, ngpt, nbnd, band_lims, gpt_dn, gpt_up, out)
        return out

    # ---- host-class helpers --------------------------------------------------------------------------
    def get_col_dry(self, vmr_h2o, plev):
        nlay, ncol = vmr_h2o.shape
        out = self.empty((nlay, ncol))
        self._c("get_col_dry", ncol, nlay, vmr_h2o, plev, out)
        return out

    def fill_gases(self, kd, vmr_by_name, col_dry):
        """col_gas(ncol,nlay,0:ngas): slot 0 = col_dry, slot i = vmr_i * col_dry
        (/root/reference/src_cuda/Gas_optics_rrtmgp.cu:392-422,1023-1028)."""
        nlay, ncol = col_dry.shape
        col_gas = self.empty((kd.ngas+1, nlay, ncol))
        vs = [vmr_by_name[name] for name in kd.gas_names]
        if len(vs) <= 32:                      # one launch for all gases
            dims = [((v.shape[1], v.shape[0]) if v.dim() == 2 else (1, 1)) for v in vs]
            self._c("fill_gases_all", ncol, nlay, len(vs), (ctypes.c_void_p * len(vs))(*[v.data_ptr() for v in vs]),
                    (ctypes.c_int * len(vs))(*[d[0] for d in dims]), (ctypes.c_int * len(vs))(*[d[1] for d in dims]), col_gas, col_dry)
            return col_gas
        vmr = self.empty((kd.ngas, nlay, ncol))
        self._c("fill_gases", ncol, nlay, ncol, nlay, kd.ngas, 0, vmr, col_dry, col_gas, col_dry)
        for i, name in enumerate(kd.gas_names, start=1):
            v = vmr_by_name[name]
            d2, d1 = (v.shape[0], v.shape[1]) if v.dim() == 2 else (1, 1)
            self._c("fill_gases", ncol, nlay, d1, d2, kd.ngas, i, vmr, v, col_gas, col_dry)
        return col_gas

    def expand_and_transpose(self, band_lims, arr_in, ngpt):
        ncol, nbnd = arr_in.shape
        out = self.empty((ngpt, ncol))
        self._c("expand_and_transpose", ncol, nbnd, band_lims, arr_in, out)
        return out

    def spread_col(self, ncol, solar_source):
        out = self.empty((solar_source.shape[0], ncol))
        self._c("spread_col", ncol, solar_source.shape[0], out, solar_source)
        return out

    def toa_source(self, ncol, solar_source, tsi_scaling):
        """spread_col + scaling_to_subset in one launch: toa_src(igpt, icol) = solar_source(igpt) * tsi_scaling(icol)."""
        out = self.empty((solar_source.shape[0], ncol))
        self._c("toa_source", ncol, solar_source.shape[0], out, solar_source, tsi_scaling)
        return out

    def scaling_to_subset(self, toa_src, tsi_scaling):
        ngpt, ncol = toa_src.shape
        self._c("scaling_to_subset", ncol, ngpt, toa_src, tsi_scaling)

    def cloud_optics_2str(self, lut, clwp, ciwp, reliq, deice, delta_scale=False):
        """delta_scale: cloud_optics followed by delta_scale_2str_k in one pass (the same bits)."""
        nlay, ncol = clwp.shape
        nbnd = lut["lut_extliq"].shape[0]
        tau = self.empty((nbnd, nlay, ncol)); ssa = self.empty((nbnd, nlay, ncol)); g = self.empty((nbnd, nlay, ncol))
        self._c("cloud_optics_2str_delta" if delta_scale else "cloud_optics_2str", ncol, nlay, nbnd, lut["nsize_liq"], lut["nsize_ice"],
                float(lut["radliq_lwr"]), float(lut["radliq_upr"]), float(lut["diamice_lwr"]), float(lut["diamice_upr"]),
                lut["lut_extliq"], lut["lut_ssaliq"], lut["lut_asyliq"], lut["lut_extice"], lut["lut_ssaice"], lut["lut_asyice"],
                clwp, ciwp, reliq, deice, tau, ssa, g)
        return tau, ssa, g

    def aerosol_optics(self, lut, aermr, rh, plev):
        """CAMS aerosol optics per band (src_cuda/Aerosol_optics.cu). aermr: aermr01..11, each (nlay, ncol) or an (nlay,)
        profile shared by all columns (read in place, not broadcast)."""
        nlay, ncol = rh.shape
        nphobic, nbnd = lut["mext_phobic"].shape
        nphilic, nhum = lut["mext_philic"].shape[:2]
        if len(aermr) != 11:
            raise ValueError("aerosol optics needs the 11 mixing ratios aermr01..aermr11")
        for m in aermr:
            if tuple(m.shape) not in ((nlay, ncol), (nlay,)):
                raise ValueError("aerosol mixing ratio must be (nlay, ncol) or (nlay,)")
        ptrs = (ctypes.c_void_p * 11)(*[m.data_ptr() for m in aermr])
        per_col = (ctypes.c_int * 11)(*[1 if m.dim() == 2 else 0 for m in aermr])
        tau = self.empty((nbnd, nlay, ncol)); ssa = self.empty((nbnd, nlay, ncol)); g = self.empty((nbnd, nlay, ncol))
        self._c("aerosol_optics", ncol, nlay, nbnd, nhum, nphobic, nphilic, ptrs, per_col, rh, plev, lut["rh_upper"],
                lut["mext_phobic"], lut["ssa_phobic"], lut["g_phobic"], lut["mext_philic"], lut["ssa_philic"], lut["g_philic"],
                tau, ssa, g)
        return tau, ssa, g

    def cloud_optics_1scl(self, lut, clwp, ciwp, reliq, deice):
        nlay, ncol = clwp.shape
        nbnd = lut["lut_extliq"].shape[0]
        tau = self.empty((nbnd, nlay, ncol))
        self._c("cloud_optics_1scl", ncol, nlay, nbnd, lut["nsize_liq"], lut["nsize_ice"],
                float(lut["radliq_lwr"]), float(lut["radliq_upr"]), float(lut["diamice_lwr"]), float(lut["diamice_upr"]),
                lut["lut_extliq"], lut["lut_ssaliq"], lut["lut_asyliq"], lut["lut_extice"], lut["lut_ssaice"], lut["lut_asyice"],
                clwp, ciwp, reliq, deice, tau)
        return tau

    def upload_lut(self, lut):
        return {k: (self.asarray(v) if isinstance(v, np.ndarray) else v) for k, v in lut.items()}

    def get_from_subset(self, ncol, nlay, nbnd, ncol_in, col_s_in, fulls, subs):
        n = len(fulls)
        PA = ctypes.c_void_p * 4
        pf = PA(*[f.data_ptr() for f in fulls] + [0]*(4-n))
        ps = PA(*[s.data_ptr() for s in subs] + [0]*(4-n))
        self._c("get_from_subset", ncol, nlay, nbnd, ncol_in, col_s_in, n, pf, ps)

    def subset_cols(self, arr, col_s, ncol_sub):
        ncol_full = arr.shape[-1]
        nrest = int(np.prod(arr.shape[:-1])) if arr.dim() > 1 else 1
        out = self.empty(tuple(arr.shape[:-1]) + (ncol_sub,))
        self._c("subset_cols", ncol_full, nrest, col_s, ncol_sub, arr, out)
        return out

    supports_null_g = True      # gas_optics_sw_fused / sw_solver_2stream accept g = None (asymmetry identically zero)

    def set_broadband_min_groups(self, n):
        """column groups needed before do_broadband takes the fused one-kernel form (default 512; 1 = always)"""
        self.lib.call("rrx_set_broadband_min_groups", int(n))

    def set_broadband_gsplit(self, n):
        self.lib.call("rrx_set_broadband_gsplit", int(n))

    def set_variant(self, lw=None, sw=None):
        if lw is not None:
            self.lib.call("rrx_set_lw_variant", int(lw))
        if sw is not None:
            self.lib.call("rrx_set_sw_variant", int(sw))
