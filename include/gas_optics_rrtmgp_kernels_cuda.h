/* Gas_optics_rrtmgp_kernels_cuda -- /root/reference/include_kernels_cuda/gas_optics_rrtmgp_kernels_cuda.h:33-132,
 * forwarding to librrx_hip.so. gas_optics_sw_fused is an addition (one-pass absorption + Rayleigh + combine). */
#ifndef GAS_OPTICS_RRTMGP_KERNELS_CUDA_H
#define GAS_OPTICS_RRTMGP_KERNELS_CUDA_H
#include "rrx_forward.h"

namespace Gas_optics_rrtmgp_kernels_cuda
{
    inline void reorder123x321(const int ni, const int nj, const int nk, const Float* arr_in, Float* arr_out)
    { RRX_CALL(rrx_reorder123x321, ni, nj, nk, arr_in, arr_out); }
    inline void reorder12x21(const int ni, const int nj, const Float* arr_in, Float* arr_out)
    { RRX_CALL(rrx_reorder12x21, ni, nj, arr_in, arr_out); }
    inline void zero_array(const int ni, const int nj, const int nk, Float* arr) { RRX_CALL(rrx_zero_array, ni, nj, nk, arr); }
    inline void zero_array(const int ni, const int nj, Float* arr) { zero_array(ni, nj, 1, arr); }
    inline void zero_array(const int ni, Float* arr) { zero_array(ni, 1, 1, arr); }

    inline void interpolation(
            const int ncol, const int nlay,
            const int ngas, const int nflav, const int neta, const int npres, const int ntemp,
            const int* flavor, const Float* press_ref_log, const Float* temp_ref,
            Float press_ref_log_delta, Float temp_ref_min, Float temp_ref_delta, Float press_ref_trop_log,
            const Float* vmr_ref, const Float* play, const Float* tlay,
            Float* col_gas, int* jtemp, Float* fmajor, Float* fminor, Float* col_mix, Bool* tropo, int* jeta, int* jpress)
    {
        RRX_CALL(rrx_interpolation, ncol, nlay, ngas, nflav, neta, npres, ntemp, flavor, press_ref_log, temp_ref,
                 press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log, vmr_ref, play, tlay, col_gas,
                 jtemp, fmajor, fminor, col_mix, tropo, jeta, jpress);
    }

    inline void combine_abs_and_rayleigh(
            const int ncol, const int nlay, const int ngpt,
            const Float* tau_local, const Float* tau_rayleigh, Float* tau, Float* ssa, Float* g)
    { RRX_CALL(rrx_combine_abs_and_rayleigh, ncol, nlay, ngpt, tau_local, tau_rayleigh, tau, ssa, g); }

    inline void compute_tau_rayleigh(
            const int ncol, const int nlay, const int nband, const int ngpt,
            const int ngas, const int nflav, const int neta, const int npres, const int ntemp,
            const int* gpoint_flavor, const int* band_lims_gpt, const Float* krayl,
            int idx_h2o, const Float* col_dry, const Float* col_gas,
            const Float* fminor, const int* jeta, const Bool* tropo, const int* jtemp, Float* tau_rayleigh)
    {
        RRX_CALL(rrx_compute_tau_rayleigh, ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp, gpoint_flavor, band_lims_gpt,
                 krayl, idx_h2o, col_dry, col_gas, fminor, jeta, tropo, jtemp, tau_rayleigh);
    }

    inline void compute_tau_absorption(
            const int ncol, const int nlay, const int nband, const int ngpt,
            const int ngas, const int nflav, const int neta, const int npres, const int ntemp,
            const int nminorlower, const int nminorklower, const int nminorupper, const int nminorkupper,
            const int idx_h2o, const int* gpoint_flavor, const int* band_lims_gpt,
            const Float* kmajor, const Float* kminor_lower, const Float* kminor_upper,
            const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper,
            const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
            const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper,
            const int* idx_minor_lower, const int* idx_minor_upper,
            const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper,
            const int* kminor_start_lower, const int* kminor_start_upper,
            const Bool* tropo, const Float* col_mix, const Float* fmajor, const Float* fminor,
            const Float* play, const Float* tlay, const Float* col_gas,
            const int* jeta, const int* jtemp, const int* jpress, Float* tau)
    {
        RRX_CALL(rrx_compute_tau_absorption, ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp,
                 nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt,
                 kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper,
                 minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper,
                 idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper,
                 kminor_start_lower, kminor_start_upper, tropo, col_mix, fmajor, fminor, play, tlay, col_gas,
                 jeta, jtemp, jpress, tau);
    }

    inline void compute_tau_absorption_set(
            const int ncol, const int nlay, const int nband, const int ngpt,
            const int ngas, const int nflav, const int neta, const int npres, const int ntemp,
            const int nminorlower, const int nminorklower, const int nminorupper, const int nminorkupper,
            const int idx_h2o, const int* gpoint_flavor, const int* band_lims_gpt,
            const Float* kmajor, const Float* kminor_lower, const Float* kminor_upper,
            const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper,
            const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
            const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper,
            const int* idx_minor_lower, const int* idx_minor_upper,
            const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper,
            const int* kminor_start_lower, const int* kminor_start_upper,
            const Bool* tropo, const Float* col_mix, const Float* fmajor, const Float* fminor,
            const Float* play, const Float* tlay, const Float* col_gas,
            const int* jeta, const int* jtemp, const int* jpress, Float* tau)
    {
        RRX_CALL(rrx_compute_tau_absorption_set, ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp,
                 nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt,
                 kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper,
                 minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper,
                 idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper,
                 kminor_start_lower, kminor_start_upper, tropo, col_mix, fmajor, fminor, play, tlay, col_gas,
                 jeta, jtemp, jpress, tau);
    }

    inline void gas_optics_sw_fused(
            const int ncol, const int nlay, const int nband, const int ngpt,
            const int ngas, const int nflav, const int neta, const int npres, const int ntemp,
            const int nminorlower, const int nminorklower, const int nminorupper, const int nminorkupper,
            const int idx_h2o, const int* gpoint_flavor, const int* band_lims_gpt,
            const Float* kmajor, const Float* kminor_lower, const Float* kminor_upper,
            const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper,
            const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
            const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper,
            const int* idx_minor_lower, const int* idx_minor_upper,
            const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper,
            const int* kminor_start_lower, const int* kminor_start_upper,
            const Bool* tropo, const Float* col_mix, const Float* fmajor, const Float* fminor,
            const Float* play, const Float* tlay, const Float* col_gas, const Float* col_dry,
            const int* jeta, const int* jtemp, const int* jpress, const Float* krayl,
            Float* tau, Float* ssa, Float* g)
    {
        RRX_CALL(rrx_gas_optics_sw_fused, ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp,
                 nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt,
                 kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper,
                 minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper,
                 idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper,
                 kminor_start_lower, kminor_start_upper, tropo, col_mix, fmajor, fminor, play, tlay, col_gas, col_dry,
                 jeta, jtemp, jpress, krayl, tau, ssa, g);
    }

    inline void compute_planck_source(
            const int ncol, const int nlay, const int nbnd, const int ngpt,
            const int nflav, const int neta, const int npres, const int ntemp, const int nPlanckTemp,
            const Float* tlay, const Float* tlev, const Float* tsfc, const int sfc_lay,
            const Float* fmajor, const int* jeta, const Bool* tropo, const int* jtemp, const int* jpress,
            const int* gpoint_bands, const int* band_lims_gpt, const Float* pfracin,
            const Float temp_ref_min, const Float totplnk_delta, const Float* totplnk, const int* gpoint_flavor,
            Float* sfc_src, Float* lay_src, Float* lev_src, Float* sfc_src_jac)
    {
        RRX_CALL(rrx_compute_planck_source, ncol, nlay, nbnd, ngpt, nflav, neta, npres, ntemp, nPlanckTemp,
                 tlay, tlev, tsfc, sfc_lay, fmajor, jeta, tropo, jtemp, jpress, gpoint_bands, band_lims_gpt, pfracin,
                 temp_ref_min, totplnk_delta, totplnk, gpoint_flavor, sfc_src, lay_src, lev_src, sfc_src_jac);
    }
}
#endif
