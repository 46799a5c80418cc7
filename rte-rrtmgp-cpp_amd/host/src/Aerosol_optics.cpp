// Aerosol_optics_gpu: /root/reference/src_cuda/Aerosol_optics.cu:266-345 (constructor, aerosol_optics); the arithmetic of its two
// kernels (:36-263) runs in one: rrx_aerosol_optics.
#include <cstdio>
#include "Aerosol_optics.h"

Aerosol_optics_gpu::Aerosol_optics_gpu(
        const Array<Float,2>& band_lims_wvn, const Array<Float,1>& rh_upper,
        const Array<Float,2>& mext_phobic, const Array<Float,2>& ssa_phobic, const Array<Float,2>& g_phobic,
        const Array<Float,3>& mext_philic, const Array<Float,3>& ssa_philic, const Array<Float,3>& g_philic) :
    Optical_props_gpu(band_lims_wvn)
{
    n_hum = rh_upper.dim(1);
    n_phobic = mext_phobic.dim(2);
    n_philic = mext_philic.dim(3);
    if (mext_phobic.dim(1) != get_nband() || mext_philic.dim(1) != get_nband() || mext_philic.dim(2) != n_hum)
        throw std::runtime_error("Aerosol_optics: table dimensions disagree with the band / humidity-class count");
    rh_upper_gpu = rh_upper;
    mext_phobic_gpu = mext_phobic; ssa_phobic_gpu = ssa_phobic; g_phobic_gpu = g_phobic;
    mext_philic_gpu = mext_philic; ssa_philic_gpu = ssa_philic; g_philic_gpu = g_philic;
}

void Aerosol_optics_gpu::aerosol_optics(
        Aerosol_concs_gpu& aerosol_concs,
        const Array_gpu<Float,2>& rh, const Array_gpu<Float,2>& plev,
        Optical_props_2str_gpu& op)
{
    const int ncol = rh.dim(1), nlay = rh.dim(2);
    const Float* mmr[11];
    int per_column[11];
    for (int i=1; i<=11; ++i)
    {
        char name[16]; std::snprintf(name, sizeof(name), "aermr%02d", i);
        const Array_gpu<Float,2>& a = aerosol_concs.get_vmr(name);
        if (a.dim(2) != nlay || (a.dim(1) != ncol && a.dim(1) != 1))
            throw std::runtime_error(std::string("Aerosol_optics: illegal dimensions of \"") + name + "\"");
        mmr[i-1] = a.ptr();
        per_column[i-1] = (a.dim(1) == 1 && ncol != 1) ? 0 : 1;
    }
    RRX_CALL(rrx_aerosol_optics, ncol, nlay, get_nband(), n_hum, n_phobic, n_philic, mmr, per_column,
             rh.ptr(), plev.ptr(), rh_upper_gpu.ptr(),
             mext_phobic_gpu.ptr(), ssa_phobic_gpu.ptr(), g_phobic_gpu.ptr(),
             mext_philic_gpu.ptr(), ssa_philic_gpu.ptr(), g_philic_gpu.ptr(),
             op.get_tau().ptr(), op.get_ssa().ptr(), op.get_g().ptr());
}
