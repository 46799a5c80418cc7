// Cloud_optics_gpu: /root/reference/src/Cloud_optics.cpp:29-69 (constructor, ice roughness category 2) and
// src_cuda/Cloud_optics.cu:181-329 (per-band LUT optics), the arithmetic runs in rrx_cloud_optics_{1scl,2str}.
#include "Cloud_optics.h"

Cloud_optics_gpu::Cloud_optics_gpu(
        const Array<Float,2>& band_lims_wvn,
        const Float radliq_lwr, const Float radliq_upr, const Float,
        const Float diamice_lwr, const Float diamice_upr, const Float,
        const Array<Float,2>& lut_extliq, const Array<Float,2>& lut_ssaliq, const Array<Float,2>& lut_asyliq,
        const Array<Float,3>& lut_extice, const Array<Float,3>& lut_ssaice, const Array<Float,3>& lut_asyice) :
    Optical_props_gpu(band_lims_wvn)
{
    liq_nsteps = lut_extliq.dim(1);
    ice_nsteps = lut_extice.dim(1);
    this->radliq_lwr = radliq_lwr; this->radliq_upr = radliq_upr;
    this->diamice_lwr = diamice_lwr; this->diamice_upr = diamice_upr;

    constexpr int icergh = 2;
    auto pick = [&](const Array<Float,3>& a)
    {
        Array<Float,2> r({a.dim(1), a.dim(2)});
        for (int ib=1; ib<=a.dim(2); ++ib)
            for (int is=1; is<=a.dim(1); ++is)
                r({is, ib}) = a({is, ib, icergh});
        return r;
    };
    lut_extliq_gpu = lut_extliq; lut_ssaliq_gpu = lut_ssaliq; lut_asyliq_gpu = lut_asyliq;
    lut_extice_gpu = pick(lut_extice); lut_ssaice_gpu = pick(lut_ssaice); lut_asyice_gpu = pick(lut_asyice);
}

void Cloud_optics_gpu::cloud_optics(
        const Array_gpu<Float,2>& clwp, const Array_gpu<Float,2>& ciwp,
        const Array_gpu<Float,2>& reliq, const Array_gpu<Float,2>& deice,
        Optical_props_2str_gpu& op, const bool delta_scale)
{
    // delta_scale: op.delta_scale() folded into the pass that produces the values (same expressions, same bits)
    if (delta_scale)
    {
        RRX_CALL(rrx_cloud_optics_2str_delta, clwp.dim(1), clwp.dim(2), this->get_nband(), liq_nsteps, ice_nsteps,
                 radliq_lwr, radliq_upr, diamice_lwr, diamice_upr,
                 lut_extliq_gpu.ptr(), lut_ssaliq_gpu.ptr(), lut_asyliq_gpu.ptr(), lut_extice_gpu.ptr(), lut_ssaice_gpu.ptr(), lut_asyice_gpu.ptr(),
                 clwp.ptr(), ciwp.ptr(), reliq.ptr(), deice.ptr(), op.get_tau().ptr(), op.get_ssa().ptr(), op.get_g().ptr());
        return;
    }
    RRX_CALL(rrx_cloud_optics_2str, clwp.dim(1), clwp.dim(2), this->get_nband(), liq_nsteps, ice_nsteps,
             radliq_lwr, radliq_upr, diamice_lwr, diamice_upr,
             lut_extliq_gpu.ptr(), lut_ssaliq_gpu.ptr(), lut_asyliq_gpu.ptr(), lut_extice_gpu.ptr(), lut_ssaice_gpu.ptr(), lut_asyice_gpu.ptr(),
             clwp.ptr(), ciwp.ptr(), reliq.ptr(), deice.ptr(), op.get_tau().ptr(), op.get_ssa().ptr(), op.get_g().ptr());
}

void Cloud_optics_gpu::cloud_optics(
        const Array_gpu<Float,2>& clwp, const Array_gpu<Float,2>& ciwp,
        const Array_gpu<Float,2>& reliq, const Array_gpu<Float,2>& deice,
        Optical_props_1scl_gpu& op)
{
    RRX_CALL(rrx_cloud_optics_1scl, clwp.dim(1), clwp.dim(2), this->get_nband(), liq_nsteps, ice_nsteps,
             radliq_lwr, radliq_upr, diamice_lwr, diamice_upr,
             lut_extliq_gpu.ptr(), lut_ssaliq_gpu.ptr(), lut_asyliq_gpu.ptr(), lut_extice_gpu.ptr(), lut_ssaice_gpu.ptr(), lut_asyice_gpu.ptr(),
             clwp.ptr(), ciwp.ptr(), reliq.ptr(), deice.ptr(), op.get_tau().ptr());
}
