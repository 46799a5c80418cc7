// Fluxes_broadband_gpu / Fluxes_byband_gpu: /root/reference/src_cuda/Fluxes.cu:27-136. A gpt_flux array whose third
// dimension is 1 is taken as already broadband (the CPU path's do_broadband convention, src/Fluxes.cpp).
#include "Fluxes.h"
#include "fluxes_kernels_cuda.h"

Fluxes_broadband_gpu::Fluxes_broadband_gpu(const int ncol, const int nlev) :
    flux_up({ncol, nlev}), flux_dn({ncol, nlev}), flux_dn_dir({ncol, nlev}), flux_net({ncol, nlev})
{}

namespace
{
    void sum_or_copy(const Array_gpu<Float,3>& gpt_flux, Array_gpu<Float,2>& flux)
    {
        const int ncol = gpt_flux.dim(1), nlev = gpt_flux.dim(2), ngpt = gpt_flux.dim(3);
        Fluxes_kernels_cuda::sum_broadband(ncol, nlev, ngpt, gpt_flux.ptr(), flux.ptr());
    }
}

void Fluxes_broadband_gpu::reduce(const Array_gpu<Float,3>& gpt_flux_up, const Array_gpu<Float,3>& gpt_flux_dn,
        const std::unique_ptr<Optical_props_arry_gpu>&, const Bool)
{
    sum_or_copy(gpt_flux_up, flux_up);
    sum_or_copy(gpt_flux_dn, flux_dn);
    Fluxes_kernels_cuda::net_broadband_precalc(flux_up.dim(1), flux_up.dim(2), flux_dn.ptr(), flux_up.ptr(), flux_net.ptr());
}

void Fluxes_broadband_gpu::reduce(const Array_gpu<Float,3>& gpt_flux_up, const Array_gpu<Float,3>& gpt_flux_dn,
        const Array_gpu<Float,3>& gpt_flux_dn_dir,
        const std::unique_ptr<Optical_props_arry_gpu>& optical_props, const Bool top_at_1)
{
    reduce(gpt_flux_up, gpt_flux_dn, optical_props, top_at_1);
    sum_or_copy(gpt_flux_dn_dir, flux_dn_dir);
}

Fluxes_byband_gpu::Fluxes_byband_gpu(const int ncol, const int nlev, const int nbnd) :
    Fluxes_broadband_gpu(ncol, nlev),
    bnd_flux_up({ncol, nlev, nbnd}), bnd_flux_dn({ncol, nlev, nbnd}), bnd_flux_dn_dir({ncol, nlev, nbnd}), bnd_flux_net({ncol, nlev, nbnd})
{}

void Fluxes_byband_gpu::reduce(const Array_gpu<Float,3>& gpt_flux_up, const Array_gpu<Float,3>& gpt_flux_dn,
        const std::unique_ptr<Optical_props_arry_gpu>& optical_props, const Bool top_at_1)
{
    const int ncol = gpt_flux_up.dim(1), nlev = gpt_flux_up.dim(2), ngpt = gpt_flux_up.dim(3);
    const int nbnd = optical_props->get_nband();
    const int* lims = optical_props->get_band_lims_gpoint_gpu().ptr();
    Fluxes_broadband_gpu::reduce(gpt_flux_up, gpt_flux_dn, optical_props, top_at_1);
    Fluxes_kernels_cuda::sum_byband(ncol, nlev, ngpt, nbnd, lims, gpt_flux_up.ptr(), bnd_flux_up.ptr());
    Fluxes_kernels_cuda::sum_byband(ncol, nlev, ngpt, nbnd, lims, gpt_flux_dn.ptr(), bnd_flux_dn.ptr());
    Fluxes_kernels_cuda::net_byband_full(ncol, nlev, ngpt, nbnd, lims, gpt_flux_dn.ptr(), gpt_flux_up.ptr(), bnd_flux_net.ptr());
}

void Fluxes_byband_gpu::reduce(const Array_gpu<Float,3>& gpt_flux_up, const Array_gpu<Float,3>& gpt_flux_dn,
        const Array_gpu<Float,3>& gpt_flux_dn_dir,
        const std::unique_ptr<Optical_props_arry_gpu>& optical_props, const Bool top_at_1)
{
    const int ncol = gpt_flux_up.dim(1), nlev = gpt_flux_up.dim(2), ngpt = gpt_flux_up.dim(3);
    reduce(gpt_flux_up, gpt_flux_dn, optical_props, top_at_1);
    Fluxes_broadband_gpu::reduce(gpt_flux_up, gpt_flux_dn, gpt_flux_dn_dir, optical_props, top_at_1);
    Fluxes_kernels_cuda::sum_byband(ncol, nlev, ngpt, optical_props->get_nband(), optical_props->get_band_lims_gpoint_gpu().ptr(),
            gpt_flux_dn_dir.ptr(), bnd_flux_dn_dir.ptr());
}
