// Optical_props_gpu family and Source_func_lw_gpu: behaviour of /root/reference/src/Optical_props.cpp:31-72 (band maps),
// src_cuda/Optical_props.cu:52-165 (containers, delta_scale, add_to) and src_cuda/Source_functions.cu.
#include "Optical_props.h"
#include "Source_functions.h"
#include "optical_props_kernels_cuda.h"
#include "gas_optics_rrtmgp_kernels_cuda.h"

Optical_props_gpu::Optical_props_gpu(const Array<Float,2>& band_lims_wvn, const Array<int,2>& band_lims_gpt)
{
    this->band2gpt = band_lims_gpt;
    this->band_lims_wvn = band_lims_wvn;
    const int nband = band_lims_gpt.dim(2);
    if (band_lims_wvn.dim(2) != nband) throw std::runtime_error("Optical_props: band limits disagree in size");
    this->gpt2band.set_dims({band_lims_gpt.max()});
    for (int ib=1; ib<=nband; ++ib)
        for (int ig=band_lims_gpt({1, ib}); ig<=band_lims_gpt({2, ib}); ++ig)
            this->gpt2band({ig}) = ib;
    this->band2gpt_gpu = this->band2gpt;
    this->gpt2band_gpu = this->gpt2band;
}

// Without g-point limits every band holds exactly one "g-point" (cloud / aerosol optics live on bands).
Optical_props_gpu::Optical_props_gpu(const Array<Float,2>& band_lims_wvn)
{
    const int nband = band_lims_wvn.dim(2);
    Array<int,2> lims({2, nband});
    for (int ib=1; ib<=nband; ++ib) { lims({1, ib}) = ib; lims({2, ib}) = ib; }
    this->band2gpt = lims;
    this->band_lims_wvn = band_lims_wvn;
    this->gpt2band.set_dims({nband});
    for (int ib=1; ib<=nband; ++ib) this->gpt2band({ib}) = ib;
    this->band2gpt_gpu = this->band2gpt;
    this->gpt2band_gpu = this->gpt2band;
}

Optical_props_1scl_gpu::Optical_props_1scl_gpu(const int ncol, const int nlay, const Optical_props_gpu& op) :
    Optical_props_arry_gpu(op), tau({ncol, nlay, this->get_ngpt()})
{}

Optical_props_2str_gpu::Optical_props_2str_gpu(const int ncol, const int nlay, const Optical_props_gpu& op) :
    Optical_props_arry_gpu(op), tau({ncol, nlay, this->get_ngpt()}), ssa({ncol, nlay, this->get_ngpt()}), g({ncol, nlay, this->get_ngpt()})
{}

void Optical_props_2str_gpu::materialize_g() const
{
    if (!g_zero) return;
    Gas_optics_rrtmgp_kernels_cuda::zero_array(g.dim(1), g.dim(2), g.dim(3), g.ptr());
    g_zero = false;
}

void Optical_props_2str_gpu::delta_scale(const Array_gpu<Float,3>& forward_frac)
{
    if (forward_frac.size() > 0) throw std::runtime_error("delta_scale with a forward fraction is not on the reference's path");
    // g == 0 everywhere (lazy form of clear-sky gas optics): f = g*g = 0, so tau, ssa and g are unchanged
    // (optical_props_kernels.cu:140-161 with g = 0 is the identity); the array behind g is not valid in this state
    if (g_zero) return;
    Optical_props_kernels_cuda::delta_scale_2str_k(get_ncol(), get_nlay(), get_ngpt(), tau.ptr(), ssa.ptr(), g.ptr());
}

void add_to(Optical_props_1scl_gpu& op_inout, const Optical_props_1scl_gpu& op_in)
{
    const int ncol = op_inout.get_ncol(), nlay = op_inout.get_nlay(), ngpt = op_inout.get_ngpt();
    if (ngpt == op_in.get_ngpt())
        Optical_props_kernels_cuda::increment_1scalar_by_1scalar(ncol, nlay, ngpt, op_inout.get_tau().ptr(), op_in.get_tau().ptr());
    else
    {
        if (op_in.get_ngpt() != op_inout.get_nband()) throw std::runtime_error("Cannot add optical properties with incompatible band - gpoint combination");
        Optical_props_kernels_cuda::inc_1scalar_by_1scalar_bybnd(ncol, nlay, ngpt, op_inout.get_tau().ptr(), op_in.get_tau().ptr(),
                op_inout.get_nband(), op_inout.get_band_lims_gpoint_gpu().ptr());
    }
}

void add_to(Optical_props_2str_gpu& op_inout, const Optical_props_2str_gpu& op_in)
{
    const int ncol = op_inout.get_ncol(), nlay = op_inout.get_nlay(), ngpt = op_inout.get_ngpt();
    if (ngpt == op_in.get_ngpt())
        Optical_props_kernels_cuda::increment_2stream_by_2stream(ncol, nlay, ngpt,
                op_inout.get_tau().ptr(), op_inout.get_ssa().ptr(), op_inout.get_g().ptr(),
                op_in.get_tau().ptr(), op_in.get_ssa().ptr(), op_in.get_g().ptr());
    else
    {
        if (op_in.get_ngpt() != op_inout.get_nband()) throw std::runtime_error("Cannot add optical properties with incompatible band - gpoint combination");
        Optical_props_kernels_cuda::inc_2stream_by_2stream_bybnd(ncol, nlay, ngpt,
                op_inout.get_tau().ptr(), op_inout.get_ssa().ptr(), op_inout.get_g().ptr(),
                op_in.get_tau().ptr(), op_in.get_ssa().ptr(), op_in.get_g().ptr(),
                op_inout.get_nband(), op_inout.get_band_lims_gpoint_gpu().ptr());
    }
}

Source_func_lw_gpu::Source_func_lw_gpu(const int n_col, const int n_lay, const Optical_props_gpu& op) :
    Optical_props_gpu(op), n_col(n_col), n_lay(n_lay),
    sfc_source({n_col, op.get_ngpt()}), sfc_source_jac({n_col, op.get_ngpt()})
{}

void Source_func_lw_gpu::ensure_full_arrays()
{
    if (lay_source.size() == 0) lay_source.set_dims({n_col, n_lay, get_ngpt()});
    if (lev_source.size() == 0) lev_source.set_dims({n_col, n_lay+1, get_ngpt()});
}

Array_gpu<Float,3>& Source_func_lw_gpu::get_planck_frac() { if (pfrac.size() == 0) pfrac.set_dims({n_col, n_lay, get_ngpt()}); return pfrac; }
Array_gpu<Float,3>& Source_func_lw_gpu::get_planck_lay()  { if (blay.size() == 0) blay.set_dims({n_col, n_lay, get_nband()}); return blay; }
Array_gpu<Float,3>& Source_func_lw_gpu::get_planck_lev()  { if (blev.size() == 0) blev.set_dims({n_col, n_lay+1, get_nband()}); return blev; }

// lay_source = pfrac*B_lay, lev_source = sqrt(pfrac*pfrac')*B_lev from the Planck-lite state (the expressions of
// /root/reference/src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:260-306)
void Source_func_lw_gpu::materialize() const
{
    Source_func_lw_gpu* self = const_cast<Source_func_lw_gpu*>(this);
    self->ensure_full_arrays();
    if (!fractions_valid) return;
    RRX_CALL(rrx_planck_sources_from_fractions, n_col, n_lay, get_ngpt(), get_gpoint_bands_gpu().ptr(),
             pfrac.ptr(), blay.ptr(), blev.ptr(), lay_source.ptr(), lev_source.ptr());
    fractions_valid = false;
}
