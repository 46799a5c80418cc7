/*
 * TEST INFRASTRUCTURE ONLY (oracle/) -- see rrtmgp_oracle.h for scope, pinning and usage rules.
 *
 * Plain scalar C++ restatement of the RTE+RRTMGP hot path. All arrays are column-major with the column
 * index fastest ("Fortran order"); index-valued arrays hold 1-based values.
 * Citations are to /root/reference files.
 */
#include <cmath>
#include <cfloat>
#include <limits>
#include <vector>
#include <algorithm>
#include <stdexcept>

#include "rrtmgp_oracle.h"

namespace
{
    inline size_t i3(int i, int j, int k, int ni, int nj) { return size_t(i) + size_t(j)*ni + size_t(k)*ni*nj; }
    const Float pi = std::acos(Float(-1.));

    // src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:2-13
    inline Float interp1d(const Float val, const Float offset, const Float delta, const int len, const Float* table)
    {
        const Float val0 = (val - offset)/delta;
        const Float frac = val0 - int(val0);
        const int idx = std::min(len-1, std::max(1, int(val0)+1));
        return table[idx-1] + frac * (table[idx] - table[idx-1]);
    }
}


extern "C"
{
// ---------------------------------------------------------------------------------------------------
// Fluxes. src_kernels_cuda/fluxes_kernels.cu:27-62; by-band semantics from
// src_kernels/mo_fluxes_byband_kernels.F90:22-71 (the CUDA by-band kernels are buggy).
// ---------------------------------------------------------------------------------------------------
void rte_sum_broadband(int* ncol, int* nlev, int* ngpt, Float* spectral_flux, Float* broadband_flux)
{
    const int nc=*ncol, nl=*nlev, ng=*ngpt;
    for (int ilev=0; ilev<nl; ++ilev)
        for (int icol=0; icol<nc; ++icol)
        {
            Float s = 0;
            for (int igpt=0; igpt<ng; ++igpt)
                s += spectral_flux[i3(icol, ilev, igpt, nc, nl)];
            broadband_flux[icol + size_t(ilev)*nc] = s;
        }
}

void rte_net_broadband_precalc(int* ncol, int* nlev, Float* flux_dn, Float* flux_up, Float* flux_net)
{
    const size_t n = size_t(*ncol) * (*nlev);
    for (size_t i=0; i<n; ++i)
        flux_net[i] = flux_dn[i] - flux_up[i];
}

void sum_byband(int* ncol, int* nlev, int* ngpt, int* nbnd, int* band_lims, Float* spectral_flux, Float* byband_flux)
{
    const int nc=*ncol, nl=*nlev, nb=*nbnd;
    for (int ibnd=0; ibnd<nb; ++ibnd)
        for (int ilev=0; ilev<nl; ++ilev)
            for (int icol=0; icol<nc; ++icol)
            {
                Float s = 0;
                for (int igpt=band_lims[2*ibnd]-1; igpt<=band_lims[2*ibnd+1]-1; ++igpt)   // inclusive, 1-based limits
                    s += spectral_flux[i3(icol, ilev, igpt, nc, nl)];
                byband_flux[i3(icol, ilev, ibnd, nc, nl)] = s;
            }
    (void)ngpt;
}

void net_byband_precalc(int* ncol, int* nlev, int* nbnd, Float* bnd_flux_dn, Float* bnd_flux_up, Float* bnd_flux_net)
{
    const size_t n = size_t(*ncol) * (*nlev) * (*nbnd);
    for (size_t i=0; i<n; ++i)
        bnd_flux_net[i] = bnd_flux_dn[i] - bnd_flux_up[i];
}

void net_byband_full(int* ncol, int* nlev, int* ngpt, int* nbnd, int* band_lims, Float* spectral_flux_dn, Float* spectral_flux_up, Float* byband_flux_net)
{
    const int nc=*ncol, nl=*nlev, nb=*nbnd;
    for (int ibnd=0; ibnd<nb; ++ibnd)
        for (int ilev=0; ilev<nl; ++ilev)
            for (int icol=0; icol<nc; ++icol)
            {
                const int g0 = band_lims[2*ibnd]-1;
                Float s = spectral_flux_dn[i3(icol, ilev, g0, nc, nl)] - spectral_flux_up[i3(icol, ilev, g0, nc, nl)];
                for (int igpt=g0+1; igpt<=band_lims[2*ibnd+1]-1; ++igpt)
                    s += spectral_flux_dn[i3(icol, ilev, igpt, nc, nl)] - spectral_flux_up[i3(icol, ilev, igpt, nc, nl)];
                byband_flux_net[i3(icol, ilev, ibnd, nc, nl)] = s;
            }
    (void)ngpt;
}

void zero_array_3D(int* ni, int* nj, int* nk, Float* array)
{
    std::fill(array, array + size_t(*ni)*(*nj)*(*nk), Float(0.));
}

void zero_array_4D(int* ni, int* nj, int* nk, int* nl, Float* array)
{
    std::fill(array, array + size_t(*ni)*(*nj)*(*nk)*(*nl), Float(0.));
}


// ---------------------------------------------------------------------------------------------------
// Gas optics.
// ---------------------------------------------------------------------------------------------------
// src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:317-395
void rrtmgp_interpolation(
        int* ncol_, int* nlay_, int* ngas_, int* nflav_, int* neta_, int* npres_, int* ntemp_,
        int* flavor, Float* press_ref_log, Float* temp_ref,
        Float* press_ref_log_delta, Float* temp_ref_min, Float* temp_ref_delta, Float* press_ref_trop_log,
        Float* vmr_ref, Float* play, Float* tlay, Float* col_gas,
        int* jtemp, Float* fmajor, Float* fminor, Float* col_mix, Bool* tropo, int* jeta, int* jpress)
{
    const int ncol=*ncol_, nlay=*nlay_, ngas=*ngas_, nflav=*nflav_, neta=*neta_, npres=*npres_, ntemp=*ntemp_;
    const Float tiny = std::numeric_limits<Float>::min();
    const size_t ncl = size_t(ncol)*nlay;

    for (int ilay=0; ilay<nlay; ++ilay)
        for (int icol=0; icol<ncol; ++icol)
        {
            const size_t idx = icol + size_t(ilay)*ncol;

            int jt = int((tlay[idx] - (*temp_ref_min - *temp_ref_delta)) / *temp_ref_delta);
            jt = std::min(ntemp-1, std::max(1, jt));
            jtemp[idx] = jt;
            const Float ftemp = (tlay[idx] - temp_ref[jt-1]) / *temp_ref_delta;

            const Float locpress = Float(1.) + (std::log(play[idx]) - press_ref_log[0]) / *press_ref_log_delta;
            const int jp = std::min(npres-1, std::max(1, int(locpress)));
            jpress[idx] = jp;
            const Float fpress = locpress - Float(jp);

            const bool in_tropo = std::log(play[idx]) > *press_ref_trop_log;
            tropo[idx] = in_tropo;
            const int itropo = in_tropo ? 0 : 1;

            for (int iflav=0; iflav<nflav; ++iflav)
            {
                const int gas1 = flavor[2*iflav];
                const int gas2 = flavor[2*iflav+1];
                const size_t cell = idx + iflav*ncl;

                for (int itemp=0; itemp<2; ++itemp)
                {
                    // vmr_ref(2, 0:ngas, ntemp)
                    const size_t vbase = itropo + size_t(jt+itemp-1) * (ngas+1) * 2;
                    const Float ratio_eta_half = vmr_ref[vbase + 2*gas1] / vmr_ref[vbase + 2*gas2];
                    const Float cg1 = col_gas[idx + gas1*ncl];
                    const Float cg2 = col_gas[idx + gas2*ncl];
                    const Float cmix = cg1 + ratio_eta_half * cg2;
                    col_mix[itemp + 2*cell] = cmix;

                    const Float eta = (cmix > Float(2.)*tiny) ? cg1 / cmix : Float(0.5);
                    const Float loceta = eta * Float(neta-1);
                    jeta[itemp + 2*cell] = std::min(int(loceta)+1, neta-1);
                    const Float feta = std::fmod(loceta, Float(1.));
                    const Float ftemp_term = Float(1-itemp) + Float(2*itemp-1)*ftemp;

                    Float* fmi = &fminor[2*(itemp + 2*cell)];
                    fmi[0] = (Float(1.)-feta) * ftemp_term;
                    fmi[1] = feta * ftemp_term;

                    Float* fma = &fmajor[4*(itemp + 2*cell)];
                    fma[0] = (Float(1.)-fpress) * fmi[0];
                    fma[1] = (Float(1.)-fpress) * fmi[1];
                    fma[2] = fpress * fmi[0];
                    fma[3] = fpress * fmi[1];
                }
            }
        }
}


namespace
{
    // src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:458-578 (one regime: idx_tropo = 1 lower, 0 upper)
    void minor_gases(
            const int ncol, const int nlay, const int ntemp, const int neta,
            const int nminor, const int idx_h2o, const int idx_tropo,
            const int* gpoint_flavor, const Float* kminor, const int* minor_limits_gpt,
            const Bool* minor_scales_with_density, const Bool* scale_by_complement,
            const int* idx_minor, const int* idx_minor_scaling, const int* kminor_start,
            const Float* play, const Float* tlay, const Float* col_gas,
            const Float* fminor, const int* jeta, const int* jtemp, const Bool* tropo, Float* tau)
    {
        const size_t ncl = size_t(ncol)*nlay;
        for (int ilay=0; ilay<nlay; ++ilay)
            for (int icol=0; icol<ncol; ++icol)
            {
                const size_t idx = icol + size_t(ilay)*ncol;
                if (tropo[idx] != idx_tropo)
                    continue;

                for (int imnr=0; imnr<nminor; ++imnr)
                {
                    Float scaling = col_gas[idx + idx_minor[imnr]*ncl];
                    if (minor_scales_with_density[imnr])
                    {
                        scaling *= Float(0.01) * play[idx] / tlay[idx];
                        if (idx_minor_scaling[imnr] > 0)
                        {
                            const Float vmr_fact = Float(1.) / col_gas[idx];
                            const Float dry_fact = Float(1.) / (Float(1.) + col_gas[idx + idx_h2o*ncl] * vmr_fact);
                            const Float x = col_gas[idx + idx_minor_scaling[imnr]*ncl] * vmr_fact * dry_fact;
                            scaling *= scale_by_complement[imnr] ? (Float(1.) - x) : x;
                        }
                    }

                    const int gpt_start = minor_limits_gpt[2*imnr]-1;
                    const int gpt_end = minor_limits_gpt[2*imnr+1];
                    const int iflav = gpoint_flavor[2*gpt_start + (1-idx_tropo)] - 1;
                    const size_t cell = idx + iflav*ncl;
                    const Float* f = &fminor[4*cell];
                    const int j0 = jeta[2*cell];
                    const int j1 = jeta[2*cell+1];
                    const int jt = jtemp[idx];
                    const int koff = kminor_start[imnr]-1;

                    for (int ig=0; ig<gpt_end-gpt_start; ++ig)
                    {
                        const size_t kb = size_t(ig+koff)*ntemp*neta;
                        const Float k =
                            f[0] * kminor[(jt-1) + (j0-1)*ntemp + kb] +
                            f[1] * kminor[(jt-1) +  j0   *ntemp + kb] +
                            f[2] * kminor[ jt    + (j1-1)*ntemp + kb] +
                            f[3] * kminor[ jt    +  j1   *ntemp + kb];
                        tau[idx + size_t(ig+gpt_start)*ncl] += k * scaling;
                    }
                }
            }
    }
}

// major: src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:398-443; sequence: ...launchers.cu:234-438.
// `tau` must be zeroed by the caller (src/Gas_optics_rrtmgp.cpp: zero_array before the call).
void rrtmgp_compute_tau_absorption(
        int* ncol_, int* nlay_, int* nband, int* ngpt_,
        int* ngas, int* nflav_, int* neta_, int* npres_, int* ntemp_,
        int* nminorlower, int* nminorklower, int* nminorupper, int* nminorkupper,
        int* idx_h2o, int* gpoint_flavor, int* band_lims_gpt,
        Float* kmajor, Float* kminor_lower, Float* kminor_upper,
        int* minor_limits_gpt_lower, int* minor_limits_gpt_upper,
        Bool* minor_scales_with_density_lower, Bool* minor_scales_with_density_upper,
        Bool* scale_by_complement_lower, Bool* scale_by_complement_upper,
        int* idx_minor_lower, int* idx_minor_upper,
        int* idx_minor_scaling_lower, int* idx_minor_scaling_upper,
        int* kminor_start_lower, int* kminor_start_upper,
        Bool* tropo, Float* col_mix, Float* fmajor, Float* fminor,
        Float* play, Float* tlay, Float* col_gas,
        int* jeta, int* jtemp, int* jpress, Float* tau)
{
    const int ncol=*ncol_, nlay=*nlay_, ngpt=*ngpt_, neta=*neta_, npres=*npres_, ntemp=*ntemp_;
    const size_t ncl = size_t(ncol)*nlay;
    const size_t s_eta = ntemp, s_prs = size_t(ntemp)*neta, s_gpt = size_t(ntemp)*neta*(npres+1);

    for (int igpt=0; igpt<ngpt; ++igpt)
        for (int ilay=0; ilay<nlay; ++ilay)
            for (int icol=0; icol<ncol; ++icol)
            {
                const size_t idx = icol + size_t(ilay)*ncol;
                const int itropo = !tropo[idx];
                const int iflav = gpoint_flavor[itropo + 2*igpt] - 1;
                const int jt = jtemp[idx];
                const int jp = jpress[idx] + itropo;
                const size_t cell = idx + iflav*ncl;
                const Float* f = &fmajor[8*cell];
                Float t = tau[idx + igpt*ncl];
                for (int i=0; i<2; ++i)
                {
                    const int je = jeta[2*cell+i];
                    const size_t b = (jt-1+i) + igpt*s_gpt;
                    t += col_mix[2*cell+i] *
                        (f[i*4+0] * kmajor[b + (je-1)*s_eta + (jp-1)*s_prs] +
                         f[i*4+1] * kmajor[b +  je   *s_eta + (jp-1)*s_prs] +
                         f[i*4+2] * kmajor[b + (je-1)*s_eta +  jp   *s_prs] +
                         f[i*4+3] * kmajor[b +  je   *s_eta +  jp   *s_prs]);
                }
                tau[idx + igpt*ncl] = t;
            }

    minor_gases(ncol, nlay, ntemp, neta, *nminorlower, *idx_h2o, 1,
            gpoint_flavor, kminor_lower, minor_limits_gpt_lower, minor_scales_with_density_lower,
            scale_by_complement_lower, idx_minor_lower, idx_minor_scaling_lower, kminor_start_lower,
            play, tlay, col_gas, fminor, jeta, jtemp, tropo, tau);
    minor_gases(ncol, nlay, ntemp, neta, *nminorupper, *idx_h2o, 0,
            gpoint_flavor, kminor_upper, minor_limits_gpt_upper, minor_scales_with_density_upper,
            scale_by_complement_upper, idx_minor_upper, idx_minor_scaling_upper, kminor_start_upper,
            play, tlay, col_gas, fminor, jeta, jtemp, tropo, tau);
    (void)nband; (void)ngas; (void)nflav_; (void)band_lims_gpt; (void)nminorklower; (void)nminorkupper;
}

// src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:94-110 : out(i,j,k) = in(k,j,i)
void reorder_123x321_kernel(int* dim1, int* dim2, int* dim3, Float* array, Float* array_out)
{
    const int ni=*dim1, nj=*dim2, nk=*dim3;
    for (int ik=0; ik<nk; ++ik)
        for (int ij=0; ij<nj; ++ij)
            for (int ii=0; ii<ni; ++ii)
                array_out[ii + size_t(ij)*ni + size_t(ik)*nj*ni] = array[ik + size_t(ij)*nk + size_t(ii)*nj*nk];
}

// Declared at rrtmgp_kernels.h:132-135 but never called by the reference; Fortran semantics: inputs are
// (ngpt,nlay,ncol), outputs (ncol,nlay,ngpt).
void combine_and_reorder_2str(int* ncol_, int* nlay_, int* ngpt_, Float* tau_local, Float* tau_rayleigh, Float* tau, Float* ssa, Float* g)
{
    const int ncol=*ncol_, nlay=*nlay_, ngpt=*ngpt_;
    const Float tiny = std::numeric_limits<Float>::min();
    for (int icol=0; icol<ncol; ++icol)
        for (int ilay=0; ilay<nlay; ++ilay)
            for (int igpt=0; igpt<ngpt; ++igpt)
            {
                const size_t in = igpt + size_t(ilay)*ngpt + size_t(icol)*ngpt*nlay;
                const size_t out = i3(icol, ilay, igpt, ncol, nlay);
                const Float t = tau_local[in] + tau_rayleigh[in];
                tau[out] = t;
                g[out] = Float(0.);
                ssa[out] = (t > Float(2.)*tiny) ? tau_rayleigh[in] / t : Float(0.);
            }
}

// CPU semantics: src/Gas_optics_rrtmgp.cpp:366-385 (threshold 2*epsilon); g = 0 as set by the
// caller (zero-initialised Optical_props_2str); GPU twin gas_optics_rrtmgp_kernels.cu:721-746.
void oracle_combine_abs_and_rayleigh(int* ncol, int* nlay, int* ngpt, Float* tau_abs, Float* tau_rayleigh, Float* tau, Float* ssa, Float* g)
{
    const size_t n = size_t(*ncol) * (*nlay) * (*ngpt);
    for (size_t i=0; i<n; ++i)
    {
        const Float t = tau_abs[i] + tau_rayleigh[i];
        ssa[i] = (t > Float(2.) * std::numeric_limits<Float>::epsilon()) ? tau_rayleigh[i] / t : Float(0.);
        tau[i] = t;
        g[i] = Float(0.);
    }
}

// src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:196-314
void rrtmgp_compute_Planck_source(
        int* ncol_, int* nlay_, int* nbnd_, int* ngpt_,
        int* nflav_, int* neta_, int* npres_, int* ntemp_, int* nPlanckTemp_,
        Float* tlay, Float* tlev, Float* tsfc, int* sfc_lay_,
        Float* fmajor, int* jeta, Bool* tropo, int* jtemp, int* jpress,
        int* gpoint_bands, int* band_lims_gpt, Float* pfracin, Float* temp_ref_min,
        Float* totplnk_delta, Float* totplnk, int* gpoint_flavor,
        Float* sfc_src, Float* lay_src, Float* lev_src, Float* sfc_src_jac)
{
    const int ncol=*ncol_, nlay=*nlay_, ngpt=*ngpt_, neta=*neta_, npres=*npres_, ntemp=*ntemp_, nPT=*nPlanckTemp_;
    const int sfc_lay = *sfc_lay_ - 1;
    const size_t ncl = size_t(ncol)*nlay;
    const size_t s_eta = ntemp, s_prs = size_t(ntemp)*neta, s_gpt = size_t(ntemp)*neta*(npres+1);
    const Float delta_Tsurf = Float(1.);

    auto pfrac_of = [&](const int icol, const int ilay, const int igpt)
    {
        const size_t idx = icol + size_t(ilay)*ncol;
        const int itropo = tropo[idx] ? 1 : 2;
        const int iflav = gpoint_flavor[(itropo-1) + 2*igpt] - 1;
        const size_t cell = idx + iflav*ncl;
        const Float* f = &fmajor[8*cell];
        const int jt = jtemp[idx];
        const int jp = jpress[idx] - 1 + itropo;         // 1-based index of the lower pressure node
        const int j0 = jeta[2*cell], j1 = jeta[2*cell+1];
        const Float* p = pfracin + igpt*s_gpt;
        return (f[0] * p[(jt-1) + (j0-1)*s_eta + (jp-1)*s_prs]
              + f[1] * p[(jt-1) +  j0   *s_eta + (jp-1)*s_prs]
              + f[2] * p[(jt-1) + (j0-1)*s_eta +  jp   *s_prs]
              + f[3] * p[(jt-1) +  j0   *s_eta +  jp   *s_prs])
             + (f[4] * p[ jt    + (j1-1)*s_eta + (jp-1)*s_prs]
              + f[5] * p[ jt    +  j1   *s_eta + (jp-1)*s_prs]
              + f[6] * p[ jt    + (j1-1)*s_eta +  jp   *s_prs]
              + f[7] * p[ jt    +  j1   *s_eta +  jp   *s_prs]);
    };

    for (int igpt=0; igpt<ngpt; ++igpt)
    {
        const Float* tp = totplnk + size_t(gpoint_bands[igpt]-1)*nPT;
        for (int ilay=0; ilay<nlay; ++ilay)
            for (int icol=0; icol<ncol; ++icol)
            {
                const size_t idx = icol + size_t(ilay)*ncol;
                const Float pfrac = pfrac_of(icol, ilay, igpt);

                lay_src[i3(icol, ilay, igpt, ncol, nlay)] = pfrac * interp1d(tlay[idx], *temp_ref_min, *totplnk_delta, nPT, tp);

                const Float b_lev = interp1d(tlev[idx], *temp_ref_min, *totplnk_delta, nPT, tp);
                if (ilay == 0)
                    lev_src[i3(icol, ilay, igpt, ncol, nlay+1)] = pfrac * b_lev;
                else
                    lev_src[i3(icol, ilay, igpt, ncol, nlay+1)] = std::sqrt(pfrac * pfrac_of(icol, ilay-1, igpt)) * b_lev;

                if (ilay == nlay-1)
                    lev_src[i3(icol, nlay, igpt, ncol, nlay+1)] =
                        pfrac * interp1d(tlev[icol + size_t(nlay)*ncol], *temp_ref_min, *totplnk_delta, nPT, tp);

                if (ilay == sfc_lay)
                {
                    const Float b1 = interp1d(tsfc[icol]              , *temp_ref_min, *totplnk_delta, nPT, tp);
                    const Float b2 = interp1d(tsfc[icol] + delta_Tsurf, *temp_ref_min, *totplnk_delta, nPT, tp);
                    sfc_src    [icol + size_t(igpt)*ncol] = pfrac * b1;
                    sfc_src_jac[icol + size_t(igpt)*ncol] = pfrac * (b2 - b1);
                }
            }
    }
    (void)nbnd_; (void)nflav_; (void)band_lims_gpt;
}

// src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:674-718
void rrtmgp_compute_tau_rayleigh(
        int* ncol_, int* nlay_, int* nband, int* ngpt_,
        int* ngas, int* nflav, int* neta_, int* npres, int* ntemp_,
        int* gpoint_flavor, int* band_lims_gpt, Float* krayl,
        int* idx_h2o, Float* col_dry, Float* col_gas,
        Float* fminor, int* jeta, Bool* tropo, int* jtemp, Float* tau_rayleigh)
{
    const int ncol=*ncol_, nlay=*nlay_, ngpt=*ngpt_, neta=*neta_, ntemp=*ntemp_;
    const size_t ncl = size_t(ncol)*nlay;
    for (int igpt=0; igpt<ngpt; ++igpt)
        for (int ilay=0; ilay<nlay; ++ilay)
            for (int icol=0; icol<ncol; ++icol)
            {
                const size_t idx = icol + size_t(ilay)*ncol;
                const int itropo = !tropo[idx];
                const int iflav = gpoint_flavor[itropo + 2*igpt] - 1;
                const size_t cell = idx + iflav*ncl;
                const Float* f = &fminor[4*cell];
                const int j0 = jeta[2*cell], j1 = jeta[2*cell+1];
                const int jt = jtemp[idx];
                const Float* k = krayl + size_t(itropo)*ntemp*neta*ngpt + size_t(igpt)*ntemp*neta;
                const Float kloc =
                    f[0] * k[(jt-1) + (j0-1)*ntemp] +
                    f[1] * k[(jt-1) +  j0   *ntemp] +
                    f[2] * k[ jt    + (j1-1)*ntemp] +
                    f[3] * k[ jt    +  j1   *ntemp];
                tau_rayleigh[idx + igpt*ncl] = kloc * (col_gas[idx + (*idx_h2o)*ncl] + col_dry[idx]);
            }
    (void)nband; (void)ngas; (void)nflav; (void)npres; (void)band_lims_gpt;
}


// ---------------------------------------------------------------------------------------------------
// Solvers.
// ---------------------------------------------------------------------------------------------------
// src_kernels_cuda/rte_solver_kernels.cu:35-193 + launchers.cu:61-286, with the Fortran-side semantics:
//  * do_broadband sums g-points into flux_*_loc(ncol,nlay+1) (src/Rte_lw.cpp:176,194-195);
//  * n_quad_angs > 1 accumulates the angles;
//  * a non-zero incident flux is treated as isotropic intensity inc/pi at every angle, so the flux at the
//    top of the domain equals inc_flux (the CUDA text halves it; all reference drivers pass none).
void rte_lw_solver_noscat(
        const int& ncol, const int& nlay, const int& ngpt, const Bool& top_at_1, const int& nmus,
        const Float* secants, const Float* weights,
        const Float* tau, const Float* lay_source, const Float* lev_source,
        const Float* sfc_emis, const Float* sfc_src, const Float* inc_flux,
        Float* flux_up, Float* flux_dn,
        const Bool& do_broadband, Float* flux_up_loc, Float* flux_dn_loc,
        const Bool& do_jacobians, const Float* sfc_src_jac, Float* flux_up_jac,
        const Bool& do_rescaling, const Float* ssa, const Float* g)
{
    if (do_rescaling)
        throw std::runtime_error("oracle: LW rescaling is not on the reference's path (src/Rte_lw.cpp:184)");
    (void)ssa; (void)g;

    const int nlev = nlay+1;
    const Float eps = std::numeric_limits<Float>::epsilon();
    const Float tau_thres = std::sqrt(std::sqrt(eps));
    std::vector<Float> trans(nlay), src_dn(nlay), src_up(nlay), up(nlev), dn(nlev), jac(nlev), tup(nlev), tdn(nlev), tjac(nlev);

    if (do_broadband)
    {
        std::fill(flux_up_loc, flux_up_loc + size_t(ncol)*nlev, Float(0.));
        std::fill(flux_dn_loc, flux_dn_loc + size_t(ncol)*nlev, Float(0.));
    }

    const int top = top_at_1 ? 0 : nlay;
    const int sfc = top_at_1 ? nlay : 0;
    const int step = top_at_1 ? 1 : -1;          // level step going down

    for (int igpt=0; igpt<ngpt; ++igpt)
        for (int icol=0; icol<ncol; ++icol)
        {
            const size_t i2 = icol + size_t(igpt)*ncol;
            std::fill(tup.begin(), tup.end(), Float(0.));
            std::fill(tdn.begin(), tdn.end(), Float(0.));
            std::fill(tjac.begin(), tjac.end(), Float(0.));

            for (int imu=0; imu<nmus; ++imu)
            {
                const Float D = secants[i2 + size_t(imu)*ncol*ngpt];
                const Float w = weights[imu];

                for (int ilay=0; ilay<nlay; ++ilay)
                {
                    const size_t il = i3(icol, ilay, igpt, ncol, nlay);
                    const Float tau_loc = tau[il] * D;
                    const Float tr = std::exp(-tau_loc);
                    const Float fact = tau_loc > tau_thres ?
                        (Float(1.) - tr) / tau_loc - tr :
                        tau_loc * (Float(.5) + tau_loc * (Float(-1./3.) + tau_loc * Float(1./8.)));
                    const Float lev_lo = lev_source[i3(icol, ilay  , igpt, ncol, nlev)];
                    const Float lev_hi = lev_source[i3(icol, ilay+1, igpt, ncol, nlev)];
                    const Float src_inc = (Float(1.) - tr) * lev_hi + Float(2.) * fact * (lay_source[il] - lev_hi);
                    const Float src_dec = (Float(1.) - tr) * lev_lo + Float(2.) * fact * (lay_source[il] - lev_lo);
                    trans[ilay] = tr;
                    src_dn[ilay] = top_at_1 ? src_inc : src_dec;
                    src_up[ilay] = top_at_1 ? src_dec : src_inc;
                }

                dn[top] = (inc_flux != nullptr) ? inc_flux[i2] / pi : Float(0.);
                for (int lev=top; lev!=sfc; lev+=step)
                {
                    const int lay = top_at_1 ? lev : lev-1;
                    dn[lev+step] = trans[lay] * dn[lev] + src_dn[lay];
                }

                up[sfc] = dn[sfc] * (Float(1.) - sfc_emis[i2]) + sfc_emis[i2] * sfc_src[i2];
                jac[sfc] = do_jacobians ? sfc_emis[i2] * sfc_src_jac[i2] : Float(0.);
                for (int lev=sfc; lev!=top; lev-=step)
                {
                    const int lay = top_at_1 ? lev-1 : lev;
                    up[lev-step] = trans[lay] * up[lev] + src_up[lay];
                    jac[lev-step] = trans[lay] * jac[lev];
                }

                for (int lev=0; lev<nlev; ++lev)
                {
                    tup[lev] += pi * w * up[lev];
                    tdn[lev] += pi * w * dn[lev];
                    tjac[lev] += pi * w * jac[lev];
                }
            }

            for (int lev=0; lev<nlev; ++lev)
            {
                if (do_broadband)
                {
                    flux_up_loc[icol + size_t(lev)*ncol] += tup[lev];
                    flux_dn_loc[icol + size_t(lev)*ncol] += tdn[lev];
                }
                else
                {
                    flux_up[i3(icol, lev, igpt, ncol, nlev)] = tup[lev];
                    flux_dn[i3(icol, lev, igpt, ncol, nlev)] = tdn[lev];
                }
                if (do_jacobians)
                    flux_up_jac[i3(icol, lev, igpt, ncol, nlev)] = tjac[lev];
            }
        }
}


// src_kernels_cuda/rte_solver_kernels.cu:196-286 (adding), :543-655 (two-stream + direct-beam source),
// launchers.cu:289-447. CPU-side semantics: mu0 is (ncol,nlay) (src/Rte_sw.cpp:160-163),
// sfc_alb_dir is indexed per g-point, has_dif_bc honoured (Q8), do_broadband sums g-points (Q5).
void rte_sw_solver_2stream(
        const int& ncol, const int& nlay, const int& ngpt, const Bool& top_at_1,
        const Float* tau, const Float* ssa, const Float* g, const Float* mu0,
        const Float* sfc_alb_dir, const Float* sfc_alb_dif, const Float* inc_flux_dir,
        Float* flux_up, Float* flux_dn, Float* flux_dir,
        const Bool& has_dif_bc, const Float* inc_flux_dif,
        const Bool& do_broadband, Float* flux_up_loc, Float* flux_dn_loc, Float* flux_dir_loc)
{
    const int nlev = nlay+1;
    const Float tmin = std::numeric_limits<Float>::epsilon();
    const Float k_min = (sizeof(Float) == 8) ? Float(1.e-12) : Float(1.e-4);
    std::vector<Float> r_dif(nlay), t_dif(nlay), s_up(nlay), s_dn(nlay), denom(nlay);
    std::vector<Float> albedo(nlev), src(nlev), up(nlev), dn(nlev), dir(nlev);

    if (do_broadband)
    {
        std::fill(flux_up_loc, flux_up_loc + size_t(ncol)*nlev, Float(0.));
        std::fill(flux_dn_loc, flux_dn_loc + size_t(ncol)*nlev, Float(0.));
        std::fill(flux_dir_loc, flux_dir_loc + size_t(ncol)*nlev, Float(0.));
    }

    const int top = top_at_1 ? 0 : nlay;
    const int sfc = top_at_1 ? nlay : 0;
    const int step = top_at_1 ? 1 : -1;

    for (int igpt=0; igpt<ngpt; ++igpt)
        for (int icol=0; icol<ncol; ++icol)
        {
            const size_t i2 = icol + size_t(igpt)*ncol;

            dir[top] = inc_flux_dir[i2] * mu0[icol + size_t(top_at_1 ? 0 : nlay-1)*ncol];
            dn[top] = (has_dif_bc && inc_flux_dif != nullptr) ? inc_flux_dif[i2] : Float(0.);

            // two-stream coefficients and direct-beam sources, marching from the top of the domain
            for (int lev=top; lev!=sfc; lev+=step)
            {
                const int lay = top_at_1 ? lev : lev-1;
                const size_t il = i3(icol, lay, igpt, ncol, nlay);
                const Float mu = mu0[icol + size_t(lay)*ncol];
                const Float mu0_inv = Float(1.)/mu;
                const Float gamma1 = (Float(8.) - ssa[il] * (Float(5.) + Float(3.) * g[il])) * Float(.25);
                const Float gamma2 = Float(3.) * (ssa[il] * (Float(1.) - g[il])) * Float(.25);
                const Float gamma3 = (Float(2.) - Float(3.) * mu * g[il]) * Float(.25);
                const Float gamma4 = Float(1.) - gamma3;
                const Float alpha1 = gamma1 * gamma4 + gamma2 * gamma3;
                const Float alpha2 = gamma1 * gamma3 + gamma2 * gamma4;
                const Float k = std::sqrt(std::max((gamma1 - gamma2) * (gamma1 + gamma2), k_min));
                const Float exp_minusktau = std::exp(-tau[il] * k);
                const Float exp_minus2ktau = exp_minusktau * exp_minusktau;
                const Float rt_term = Float(1.) / (k * (Float(1.) + exp_minus2ktau) + gamma1 * (Float(1.) - exp_minus2ktau));
                r_dif[lay] = rt_term * gamma2 * (Float(1.) - exp_minus2ktau);
                t_dif[lay] = rt_term * Float(2.) * k * exp_minusktau;
                const Float t_noscat = std::exp(-tau[il] * mu0_inv);
                const Float k_mu = k * mu;
                const Float k_gamma3 = k * gamma3;
                const Float k_gamma4 = k * gamma4;
                const Float fact = (std::abs(Float(1.) - k_mu*k_mu) > tmin) ? Float(1.) - k_mu*k_mu : tmin;
                const Float rt_term2 = ssa[il] * rt_term / fact;
                Float r_dir = rt_term2 * ((Float(1.) - k_mu) * (alpha2 + k_gamma3) -
                                          (Float(1.) + k_mu) * (alpha2 - k_gamma3) * exp_minus2ktau -
                                          Float(2.) * (k_gamma3 - alpha2 * k_mu) * exp_minusktau * t_noscat);
                Float t_dir = -rt_term2 * ((Float(1.) + k_mu) * (alpha1 + k_gamma4) * t_noscat -
                                           (Float(1.) - k_mu) * (alpha1 - k_gamma4) * exp_minus2ktau * t_noscat -
                                           Float(2.) * (k_gamma4 + alpha1 * k_mu) * exp_minusktau);
                r_dir = std::max(tmin, std::min(r_dir, Float(1.) - t_noscat));
                t_dir = std::max(tmin, std::min(t_dir, Float(1.) - t_noscat - r_dir));

                s_up[lay] = r_dir * dir[lev];
                s_dn[lay] = t_dir * dir[lev];
                dir[lev+step] = t_noscat * dir[lev];
            }

            // adding: from the surface to the top
            albedo[sfc] = sfc_alb_dif[i2];
            src[sfc] = dir[sfc] * sfc_alb_dir[i2];
            for (int lev=sfc; lev!=top; lev-=step)
            {
                const int lay = top_at_1 ? lev-1 : lev;
                denom[lay] = Float(1.)/(Float(1.) - r_dif[lay] * albedo[lev]);
                albedo[lev-step] = r_dif[lay] + t_dif[lay] * t_dif[lay] * albedo[lev] * denom[lay];
                src[lev-step] = s_up[lay] + t_dif[lay] * denom[lay] * (src[lev] + albedo[lev] * s_dn[lay]);
            }

            up[top] = dn[top] * albedo[top] + src[top];
            for (int lev=top; lev!=sfc; lev+=step)
            {
                const int lay = top_at_1 ? lev : lev-1;
                dn[lev+step] = (t_dif[lay] * dn[lev] + r_dif[lay] * src[lev+step] + s_dn[lay]) * denom[lay];
                up[lev+step] = dn[lev+step] * albedo[lev+step] + src[lev+step];
            }

            for (int lev=0; lev<nlev; ++lev)
            {
                const Float dn_tot = dn[lev] + dir[lev];
                if (do_broadband)
                {
                    flux_up_loc[icol + size_t(lev)*ncol] += up[lev];
                    flux_dn_loc[icol + size_t(lev)*ncol] += dn_tot;
                    flux_dir_loc[icol + size_t(lev)*ncol] += dir[lev];
                }
                else
                {
                    flux_up[i3(icol, lev, igpt, ncol, nlev)] = up[lev];
                    flux_dn[i3(icol, lev, igpt, ncol, nlev)] = dn_tot;
                    flux_dir[i3(icol, lev, igpt, ncol, nlev)] = dir[lev];
                }
            }
        }
}


// ---------------------------------------------------------------------------------------------------
// Optical properties. src_kernels_cuda/optical_props_kernels.cu:31-161, eps = 3*tiny (launchers :80).
// ---------------------------------------------------------------------------------------------------
void rte_increment_1scalar_by_1scalar(int* ncol, int* nlay, int* ngpt, Float* tau_inout, Float* tau_in)
{
    const size_t n = size_t(*ncol) * (*nlay) * (*ngpt);
    for (size_t i=0; i<n; ++i)
        tau_inout[i] = tau_inout[i] + tau_in[i];
}

namespace
{
    inline void inc_2str(Float& tau1, Float& ssa1, Float& g1, const Float tau2, const Float ssa2, const Float g2, const Float eps)
    {
        const Float tau12 = tau1 + tau2;
        const Float tauscat12 = (tau1 * ssa1) + (tau2 * ssa2);
        g1 = ((tau1 * ssa1 * g1) + (tau2 * ssa2 * g2)) / std::max(tauscat12, eps);
        ssa1 = tauscat12 / std::max(eps, tau12);
        tau1 = tau12;
    }
}

void rte_increment_2stream_by_2stream(int* ncol, int* nlay, int* ngpt, Float* tau_inout, Float* ssa_inout, Float* g_inout, Float* tau_in, Float* ssa_in, Float* g_in)
{
    const Float eps = std::numeric_limits<Float>::min() * Float(3.);
    const size_t n = size_t(*ncol) * (*nlay) * (*ngpt);
    for (size_t i=0; i<n; ++i)
        inc_2str(tau_inout[i], ssa_inout[i], g_inout[i], tau_in[i], ssa_in[i], g_in[i], eps);
}

void rte_inc_1scalar_by_1scalar_bybnd(int* ncol, int* nlay, int* ngpt, Float* tau_inout, Float* tau_in, int* nbnd, int* band_lims_gpoint)
{
    const size_t ncl = size_t(*ncol) * (*nlay);
    for (int ibnd=0; ibnd<*nbnd; ++ibnd)
        for (int igpt=band_lims_gpoint[2*ibnd]-1; igpt<=band_lims_gpoint[2*ibnd+1]-1; ++igpt)
            for (size_t i=0; i<ncl; ++i)
                tau_inout[i + igpt*ncl] = tau_inout[i + igpt*ncl] + tau_in[i + ibnd*ncl];
    (void)ngpt;
}

void rte_inc_2stream_by_2stream_bybnd(int* ncol, int* nlay, int* ngpt, Float* tau_inout, Float* ssa_inout, Float* g_inout, Float* tau_in, Float* ssa_in, Float* g_in, int* nbnd, int* band_lims_gpoint)
{
    const Float eps = std::numeric_limits<Float>::min() * Float(3.);
    const size_t ncl = size_t(*ncol) * (*nlay);
    for (int ibnd=0; ibnd<*nbnd; ++ibnd)
        for (int igpt=band_lims_gpoint[2*ibnd]-1; igpt<=band_lims_gpoint[2*ibnd+1]-1; ++igpt)
            for (size_t i=0; i<ncl; ++i)
                inc_2str(tau_inout[i + igpt*ncl], ssa_inout[i + igpt*ncl], g_inout[i + igpt*ncl],
                         tau_in[i + ibnd*ncl], ssa_in[i + ibnd*ncl], g_in[i + ibnd*ncl], eps);
    (void)ngpt;
}

void rte_delta_scale_2str_k(int* ncol, int* nlay, int* ngpt, Float* tau, Float* ssa, Float* g)
{
    const Float eps = std::numeric_limits<Float>::min() * Float(3.);
    const size_t n = size_t(*ncol) * (*nlay) * (*ngpt);
    for (size_t i=0; i<n; ++i)
    {
        const Float f = g[i] * g[i];
        const Float wf = ssa[i] * f;
        tau[i] *= (Float(1.) - wf);
        ssa[i] = (ssa[i] - wf) / std::max(eps, Float(1.) - wf);
        g[i] = (g[i] - f) / std::max(eps, Float(1.) - f);
    }
}


// ---------------------------------------------------------------------------------------------------
// Host-class arithmetic outside the Fortran boundary.
// ---------------------------------------------------------------------------------------------------
// src/Gas_optics_rrtmgp.cpp:764-792
void oracle_get_col_dry(int* ncol_, int* nlay_, Float* vmr_h2o, Float* plev, Float* col_dry)
{
    const int ncol=*ncol_, nlay=*nlay_;
    constexpr Float g0 = 9.80665;
    constexpr Float avogad = 6.02214076e23;
    constexpr Float m_dry = 0.028964;
    constexpr Float m_h2o = 0.018016;
    for (int ilay=0; ilay<nlay; ++ilay)
        for (int icol=0; icol<ncol; ++icol)
        {
            const size_t idx = icol + size_t(ilay)*ncol;
            const Float delta_plev = std::abs(plev[idx] - plev[idx + ncol]);
            const Float m_air = (m_dry + m_h2o * vmr_h2o[idx]) / (1. + vmr_h2o[idx]);
            Float cd = Float(10.) * delta_plev * avogad / (Float(1000.)*m_air*Float(100.)*g0);
            cd /= (Float(1.) + vmr_h2o[idx]);
            col_dry[idx] = cd;
        }
}

// src/Rte_lw.cpp:70-93 / src_cuda/Rte_lw.cu:37-56 : in (nbnd,ncol) -> out (ncol,ngpt)
void oracle_expand_and_transpose(int* ncol_, int* nbnd_, int* ngpt, int* band_lims_gpt, Float* arr_in, Float* arr_out)
{
    const int ncol=*ncol_, nbnd=*nbnd_;
    for (int ibnd=0; ibnd<nbnd; ++ibnd)
        for (int igpt=band_lims_gpt[2*ibnd]-1; igpt<=band_lims_gpt[2*ibnd+1]-1; ++igpt)
            for (int icol=0; icol<ncol; ++icol)
                arr_out[icol + size_t(igpt)*ncol] = arr_in[ibnd + size_t(icol)*nbnd];
    (void)ngpt;
}

namespace
{
    // src/Cloud_optics.cpp:72-107
    inline void cloud_from_table(
            const Float cwp, const Float re, const int nsteps, const Float step_size, const Float offset,
            const Float* tau_table, const Float* ssa_table, const Float* asy_table,
            Float& tau, Float& taussa, Float& taussag)
    {
        if (cwp > Float(0.))
        {
            const int index = std::min(static_cast<int>((re - offset) / step_size)+1, nsteps-1);
            const Float fint = (re - offset) / step_size - (index-1);
            tau = cwp * (tau_table[index-1] + fint * (tau_table[index] - tau_table[index-1]));
            taussa = tau * (ssa_table[index-1] + fint * (ssa_table[index] - ssa_table[index-1]));
            taussag = taussa * (asy_table[index-1] + fint * (asy_table[index] - asy_table[index-1]));
        }
        else
        {
            tau = taussa = taussag = Float(0.);
        }
    }
}

// src/Cloud_optics.cpp:111-172. LUTs are (nsize, nbnd), ice already reduced to roughness category 2 (:61-68).
void oracle_cloud_optics_2str(
        int* ncol_, int* nlay_, int* nbnd_, int* nsize_liq, int* nsize_ice,
        Float* radliq_lwr, Float* radliq_upr, Float* diamice_lwr, Float* diamice_upr,
        Float* lut_extliq, Float* lut_ssaliq, Float* lut_asyliq,
        Float* lut_extice, Float* lut_ssaice, Float* lut_asyice,
        Float* clwp, Float* ciwp, Float* reliq, Float* deice,
        Float* tau, Float* ssa, Float* g)
{
    const int ncol=*ncol_, nlay=*nlay_, nbnd=*nbnd_;
    const Float liq_step = (*radliq_upr - *radliq_lwr) / (*nsize_liq - Float(1.));
    const Float ice_step = (*diamice_upr - *diamice_lwr) / (*nsize_ice - Float(1.));
    const Float eps = std::numeric_limits<Float>::epsilon();
    const size_t ncl = size_t(ncol)*nlay;
    for (int ibnd=0; ibnd<nbnd; ++ibnd)
        for (size_t i=0; i<ncl; ++i)
        {
            Float lt, lts, ltsg, it, its, itsg;
            cloud_from_table(clwp[i], reliq[i], *nsize_liq, liq_step, *radliq_lwr,
                    lut_extliq + size_t(ibnd)*(*nsize_liq), lut_ssaliq + size_t(ibnd)*(*nsize_liq), lut_asyliq + size_t(ibnd)*(*nsize_liq), lt, lts, ltsg);
            cloud_from_table(ciwp[i], deice[i], *nsize_ice, ice_step, *diamice_lwr,
                    lut_extice + size_t(ibnd)*(*nsize_ice), lut_ssaice + size_t(ibnd)*(*nsize_ice), lut_asyice + size_t(ibnd)*(*nsize_ice), it, its, itsg);
            const Float t = lt + it, ts = lts + its, tsg = ltsg + itsg;
            tau[i + ibnd*ncl] = t;
            ssa[i + ibnd*ncl] = ts / std::max(t, eps);
            g  [i + ibnd*ncl] = tsg / std::max(ts, eps);
        }
}

// src/Cloud_optics.cpp:176-232
void oracle_cloud_optics_1scl(
        int* ncol_, int* nlay_, int* nbnd_, int* nsize_liq, int* nsize_ice,
        Float* radliq_lwr, Float* radliq_upr, Float* diamice_lwr, Float* diamice_upr,
        Float* lut_extliq, Float* lut_ssaliq, Float* lut_asyliq,
        Float* lut_extice, Float* lut_ssaice, Float* lut_asyice,
        Float* clwp, Float* ciwp, Float* reliq, Float* deice,
        Float* tau)
{
    const int ncol=*ncol_, nlay=*nlay_, nbnd=*nbnd_;
    const Float liq_step = (*radliq_upr - *radliq_lwr) / (*nsize_liq - Float(1.));
    const Float ice_step = (*diamice_upr - *diamice_lwr) / (*nsize_ice - Float(1.));
    const size_t ncl = size_t(ncol)*nlay;
    for (int ibnd=0; ibnd<nbnd; ++ibnd)
        for (size_t i=0; i<ncl; ++i)
        {
            Float lt, lts, ltsg, it, its, itsg;
            cloud_from_table(clwp[i], reliq[i], *nsize_liq, liq_step, *radliq_lwr,
                    lut_extliq + size_t(ibnd)*(*nsize_liq), lut_ssaliq + size_t(ibnd)*(*nsize_liq), lut_asyliq + size_t(ibnd)*(*nsize_liq), lt, lts, ltsg);
            cloud_from_table(ciwp[i], deice[i], *nsize_ice, ice_step, *diamice_lwr,
                    lut_extice + size_t(ibnd)*(*nsize_ice), lut_ssaice + size_t(ibnd)*(*nsize_ice), lut_asyice + size_t(ibnd)*(*nsize_ice), it, its, itsg);
            tau[i + ibnd*ncl] = (lt - lts) + (it - its);
        }
}

// src/Aerosol_optics.cpp:24-36 (rh_class), :38-157 (compute_all_from_table), :176-224 (finalisation). Species in the order of
// :56 (SS1 SS2 SS3 DU1 DU2 DU3 OM1 OM2 BC1 BC2 SU); tables: hydrophobic (nbnd, nphobic), hydrophilic (nbnd, nhum, nphilic),
// band fastest. Two deliberate differences from that text, both shared with the GPU text (src_cuda/Aerosol_optics.cu:65):
// dp is taken as |dp| (the CPU text gives negative optical depths for a top-first ordering), and a humidity above the last
// class bound uses the last class (the reference reads past the table).
void oracle_aerosol_optics_2str(
        int* ncol_, int* nlay_, int* nbnd_, int* nhum_,
        Float* aermr01, Float* aermr02, Float* aermr03, Float* aermr04, Float* aermr05, Float* aermr06,
        Float* aermr07, Float* aermr08, Float* aermr09, Float* aermr10, Float* aermr11,
        Float* rh, Float* plev, Float* rh_classes,
        Float* mext_phobic, Float* ssa_phobic, Float* g_phobic,
        Float* mext_philic, Float* ssa_philic, Float* g_philic,
        Float* tau, Float* ssa, Float* g)
{
    const int ncol=*ncol_, nlay=*nlay_, nbnd=*nbnd_, nhum=*nhum_;
    const Float eps = std::numeric_limits<Float>::epsilon();
    const size_t ncl = size_t(ncol)*nlay;
    // {mixing ratio, hydrophilic?, 1-based table column} per species, in accumulation order
    struct Species { const Float* mmr; bool philic; int col; };
    const Species list[11] = {
        {aermr01, true, 1}, {aermr02, true, 2}, {aermr03, true, 3},
        {aermr04, false, 1}, {aermr05, false, 8}, {aermr06, false, 6},
        {aermr08, false, 10}, {aermr07, true, 4},
        {aermr09, false, 11}, {aermr10, false, 11},
        {aermr11, true, 5}};
    for (int ibnd=0; ibnd<nbnd; ++ibnd)
        for (size_t i=0; i<ncl; ++i)
        {
            const Float dpg = std::abs(plev[i] - plev[i + ncol]) / Float(9.81);
            int ihum = 1;
            while (ihum < nhum && rh_classes[ihum-1] < rh[i]) ++ihum;
            Float tau_local = 0, taussa_local = 0, taussag_local = 0;
            for (const Species& sp : list)
            {
                const size_t k = sp.philic ? size_t(ibnd) + size_t(ihum-1)*nbnd + size_t(sp.col-1)*nbnd*nhum
                                           : size_t(ibnd) + size_t(sp.col-1)*nbnd;
                const Float mext = sp.philic ? mext_philic[k] : mext_phobic[k];
                const Float w    = sp.philic ? ssa_philic[k]  : ssa_phobic[k];
                const Float asy  = sp.philic ? g_philic[k]    : g_phobic[k];
                const Float local_od = sp.mmr[i] * dpg * mext;
                tau_local += local_od;
                taussa_local += local_od * w;
                taussag_local += local_od * w * asy;
            }
            tau[i + ibnd*ncl] = tau_local;
            ssa[i + ibnd*ncl] = taussa_local / std::max(tau_local, eps);
            g  [i + ibnd*ncl] = taussag_local / std::max(taussa_local, eps);
        }
}
}
