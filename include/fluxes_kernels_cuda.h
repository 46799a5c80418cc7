/* Fluxes_kernels_cuda -- /root/reference/include_kernels_cuda/fluxes_kernels_cuda.h:33-51. The by-band functions follow
 * the Fortran semantics (src_kernels/mo_fluxes_byband_kernels.F90): spectral input, 1-based inclusive band limits. */
#ifndef FLUXES_KERNELS_CUDA_H
#define FLUXES_KERNELS_CUDA_H
#include "rrx_forward.h"

namespace Fluxes_kernels_cuda
{
    inline void sum_broadband(int ncol, int nlev, int ngpt, const Float* gpt_flux, Float* flux)
    { RRX_CALL(rrx_sum_broadband, ncol, nlev, ngpt, gpt_flux, flux); }
    inline void net_broadband_precalc(int ncol, int nlev, const Float* broadband_flux_dn, const Float* broadband_flux_up, Float* broadband_flux_net)
    { RRX_CALL(rrx_net_broadband_precalc, ncol, nlev, broadband_flux_dn, broadband_flux_up, broadband_flux_net); }
    inline void sum_byband(int ncol, int nlev, int ngpt, int nbnd, const int* band_lims, const Float* gpt_flux, Float* bnd_flux)
    { RRX_CALL(rrx_sum_byband, ncol, nlev, ngpt, nbnd, band_lims, gpt_flux, bnd_flux); }
    inline void net_byband_full(int ncol, int nlev, int ngpt, int nbnd, const int* band_lims, const Float* gpt_flux_dn, const Float* gpt_flux_up, Float* bnd_flux_net)
    { RRX_CALL(rrx_net_byband_full, ncol, nlev, ngpt, nbnd, band_lims, gpt_flux_dn, gpt_flux_up, bnd_flux_net); }
}
#endif
