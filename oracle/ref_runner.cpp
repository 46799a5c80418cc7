/*
 * TEST INFRASTRUCTURE ONLY (oracle/): never linked into, imported by or called from the product path.
 *
 * oracle/_ref/libref_{dp,sp}.so : the reference's own CUDA kernel text, #included IN PLACE from
 * /root/reference/src_kernels_cuda/*.cu (never copied), executed sequentially on the host through
 * oracle/ref_shim/cuda_on_host.h. Built only in the build container (see oracle/Makefile, target `ref`);
 * the GPU box only ever sees the prebuilt .so.
 *
 * Each ref_* entry point replays the kernel sequence of the matching reference launcher:
 *   ref_lw_solver_noscat        <- src_kernels_cuda/rte_solver_kernels_launchers.cu:61-286
 *   ref_sw_solver_2stream       <- src_kernels_cuda/rte_solver_kernels_launchers.cu:289-447
 *   ref_interpolation           <- src_kernels_cuda/gas_optics_rrtmgp_kernels_launchers.cu:91-125
 *   ref_compute_tau_absorption  <- ...launchers.cu:234-438   (major, minor lower, minor upper)
 *   ref_compute_tau_rayleigh    <- ...launchers.cu:168-221
 *   ref_combine_abs_and_rayleigh<- ...launchers.cu:128-165
 *   ref_compute_planck_source   <- ...launchers.cu:441-521
 *   ref_increment_* / ref_inc_*_bybnd / ref_delta_scale_2str_k <- optical_props_kernels_launchers.cu:42-185
 *   ref_sum_broadband / ref_net_broadband_precalc <- fluxes_kernels_launchers.cu
 * Scalars are passed by value and arrays as raw pointers, exactly like include_kernels_cuda/*.h.
 */
#include <vector>
#include <limits>
#include <stdexcept>

#include "ref_shim/cuda_on_host.h"
#include "types.h"   // the reference's include/types.h (Float, Bool)

namespace ref_rte
{
    #include "rte_solver_kernels.cu"
}
namespace ref_gas
{
    #include "gas_optics_rrtmgp_kernels.cu"
}
namespace ref_opt
{
    #include "optical_props_kernels.cu"
}
namespace ref_flx
{
    #include "fluxes_kernels.cu"
}

extern "C"
{
int ref_sizeof_float() { return (int)sizeof(Float); }
int ref_sizeof_bool() { return (int)sizeof(Bool); }


void ref_lw_secants_array(
        const int ncol, const int ngpt, const int n_gauss_quad, const int max_gauss_pts,
        const Float* gauss_Ds, Float* secants)
{
    host_launch(dim3(ncol, ngpt, n_gauss_quad), ref_rte::lw_secants_array_kernel,
            ncol, ngpt, n_gauss_quad, max_gauss_pts, gauss_Ds, secants);
}


void ref_lw_solver_noscat(
        const int ncol, const int nlay, const int ngpt, const Bool top_at_1, const int nmus,
        const Float* secants, const Float* weights,
        const Float* tau, const Float* lay_source, const Float* lev_source,
        const Float* sfc_emis, const Float* sfc_src,
        const Float* inc_flux,
        Float* flux_up, Float* flux_dn,
        const Float* sfc_src_jac, Float* flux_up_jac)
{
    if (nmus != 1)
        throw std::runtime_error("reference GPU path implements nmus == 1 only");

    const Float eps = std::numeric_limits<Float>::epsilon();
    const size_t flx_size = size_t(ncol)*(nlay+1)*ngpt;
    const size_t opt_size = size_t(ncol)*nlay*ngpt;
    const size_t sfc_size = size_t(ncol)*ngpt;

    std::vector<Float> source_sfc(sfc_size), source_sfc_jac(sfc_size), sfc_albedo(sfc_size);
    std::vector<Float> tau_loc(opt_size), trans(opt_size), source_dn(opt_size), source_up(opt_size);
    std::vector<Float> radn_dn(flx_size);

    if (inc_flux == nullptr)
        host_launch(dim3(ncol, ngpt), static_cast<void(*)(const int, const int, const int, const Bool, Float*)>(ref_rte::apply_BC_kernel),
                ncol, nlay, ngpt, top_at_1, flux_dn);
    else
        host_launch(dim3(ncol, ngpt), static_cast<void(*)(const int, const int, const int, const Bool, const Float*, Float*)>(ref_rte::apply_BC_kernel),
                ncol, nlay, ngpt, top_at_1, inc_flux, flux_dn);

    host_launch(dim3(ncol, nlay, ngpt), ref_rte::lw_solver_noscat_step_1_kernel,
            ncol, nlay, ngpt, eps, top_at_1, secants, weights, tau, lay_source, lev_source,
            sfc_emis, sfc_src, flux_up, flux_dn, sfc_src_jac, flux_up_jac,
            tau_loc.data(), trans.data(), source_dn.data(), source_up.data(),
            source_sfc.data(), sfc_albedo.data(), source_sfc_jac.data());

    host_launch(dim3(ncol, ngpt), ref_rte::lw_solver_noscat_step_2_kernel,
            ncol, nlay, ngpt, eps, top_at_1, secants, weights, tau, lay_source, lev_source,
            sfc_emis, sfc_src, flux_up, flux_dn, sfc_src_jac, flux_up_jac,
            tau_loc.data(), trans.data(), source_dn.data(), source_up.data(),
            source_sfc.data(), sfc_albedo.data(), source_sfc_jac.data());

    host_launch(dim3(ncol, nlay+1, ngpt), ref_rte::lw_solver_noscat_step_3_kernel,
            ncol, nlay, ngpt, eps, top_at_1, secants, weights, tau, lay_source, lev_source,
            sfc_emis, sfc_src, flux_up, flux_dn, sfc_src_jac, flux_up_jac,
            tau_loc.data(), trans.data(), source_dn.data(), source_up.data(),
            source_sfc.data(), sfc_albedo.data(), source_sfc_jac.data());

    const int top_level = top_at_1 ? 0 : nlay;
    host_launch(dim3(ncol, ngpt), ref_rte::apply_BC_kernel_lw,
            top_level, ncol, nlay, ngpt, top_at_1, flux_dn, radn_dn.data());
}


void ref_sw_solver_2stream(
        const int ncol, const int nlay, const int ngpt, const Bool top_at_1,
        const Float* tau, const Float* ssa, const Float* g,
        const Float* mu0,
        const Float* sfc_alb_dir, const Float* sfc_alb_dif,
        const Float* inc_flux_dir,
        Float* flux_up, Float* flux_dn, Float* flux_dir,
        const Float* inc_flux_dif)
{
    const size_t opt_size = size_t(ncol)*nlay*ngpt;
    const size_t alb_size = size_t(ncol)*ngpt;
    const size_t flx_size = size_t(ncol)*(nlay+1)*ngpt;

    std::vector<Float> r_dif(opt_size), t_dif(opt_size), source_up(opt_size), source_dn(opt_size), denom(opt_size);
    std::vector<Float> source_sfc(alb_size), albedo(flx_size), src(flx_size);

    host_launch(dim3(ncol, ngpt), static_cast<void(*)(const int, const int, const int, const Bool, const Float*, const Float*, Float*)>(ref_rte::apply_BC_kernel),
            ncol, nlay, ngpt, top_at_1, inc_flux_dir, mu0, flux_dir);
    if (inc_flux_dif == nullptr)
        host_launch(dim3(ncol, ngpt), static_cast<void(*)(const int, const int, const int, const Bool, Float*)>(ref_rte::apply_BC_kernel),
                ncol, nlay, ngpt, top_at_1, flux_dn);
    else
        host_launch(dim3(ncol, ngpt), static_cast<void(*)(const int, const int, const int, const Bool, const Float*, Float*)>(ref_rte::apply_BC_kernel),
                ncol, nlay, ngpt, top_at_1, inc_flux_dif, flux_dn);

    if (top_at_1)
    {
        host_launch(dim3(ncol, ngpt), ref_rte::sw_source_2stream_kernel<1>,
                ncol, nlay, ngpt, tau, ssa, g, mu0, r_dif.data(), t_dif.data(),
                sfc_alb_dir, source_up.data(), source_dn.data(), source_sfc.data(), flux_dir);
        host_launch(dim3(ncol, ngpt), ref_rte::sw_adding_kernel<1>,
                ncol, nlay, ngpt, top_at_1, sfc_alb_dif, r_dif.data(), t_dif.data(),
                source_dn.data(), source_up.data(), source_sfc.data(),
                flux_up, flux_dn, flux_dir, albedo.data(), src.data(), denom.data());
    }
    else
    {
        host_launch(dim3(ncol, ngpt), ref_rte::sw_source_2stream_kernel<0>,
                ncol, nlay, ngpt, tau, ssa, g, mu0, r_dif.data(), t_dif.data(),
                sfc_alb_dir, source_up.data(), source_dn.data(), source_sfc.data(), flux_dir);
        host_launch(dim3(ncol, ngpt), ref_rte::sw_adding_kernel<0>,
                ncol, nlay, ngpt, top_at_1, sfc_alb_dif, r_dif.data(), t_dif.data(),
                source_dn.data(), source_up.data(), source_sfc.data(),
                flux_up, flux_dn, flux_dir, albedo.data(), src.data(), denom.data());
    }
}


void ref_interpolation(
        const int ncol, const int nlay,
        const int ngas, const int nflav, const int neta, const int npres, const int ntemp,
        const int* flavor, const Float* press_ref_log, const Float* temp_ref,
        Float press_ref_log_delta, Float temp_ref_min, Float temp_ref_delta, Float press_ref_trop_log,
        const Float* vmr_ref, const Float* play, const Float* tlay,
        Float* col_gas, int* jtemp, Float* fmajor, Float* fminor, Float* col_mix,
        Bool* tropo, int* jeta, int* jpress)
{
    const Float tmin = std::numeric_limits<Float>::min();
    host_launch(dim3(ncol, nlay, nflav), ref_gas::interpolation_kernel,
            ncol, nlay, ngas, nflav, neta, npres, ntemp, tmin,
            flavor, press_ref_log, temp_ref, press_ref_log_delta, temp_ref_min, temp_ref_delta,
            press_ref_trop_log, vmr_ref, play, tlay, col_gas, jtemp, fmajor, fminor, col_mix,
            tropo, jeta, jpress);
}


void ref_compute_tau_absorption(
        const int ncol, const int nlay, const int nband, const int ngpt,
        const int ngas, const int nflav, const int neta, const int npres, const int ntemp,
        const int nminorlower, const int nminorklower,
        const int nminorupper, const int nminorkupper,
        const int idx_h2o,
        const int* gpoint_flavor, const int* band_lims_gpt,
        const Float* kmajor, const Float* kminor_lower, const Float* kminor_upper,
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper,
        const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
        const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper,
        const int* idx_minor_lower, const int* idx_minor_upper,
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper,
        const int* kminor_start_lower, const int* kminor_start_upper,
        const Bool* tropo,
        const Float* col_mix, const Float* fmajor, const Float* fminor,
        const Float* play, const Float* tlay, const Float* col_gas,
        const int* jeta, const int* jtemp, const int* jpress,
        Float* tau)
{
    host_launch(dim3(ngpt, nlay, ncol), ref_gas::gas_optical_depths_major_kernel,
            ncol, nlay, nband, ngpt, nflav, neta, npres, ntemp,
            gpoint_flavor, band_lims_gpt, kmajor, col_mix, fmajor, jeta, tropo, jtemp, jpress, tau);

    int idx_tropo = 1;
    host_launch(dim3(1, nlay, ncol), ref_gas::gas_optical_depths_minor_kernel<1,1,1>,
            ncol, nlay, ngpt, ngas, nflav, ntemp, neta, nminorlower, nminorklower, idx_h2o, idx_tropo,
            gpoint_flavor, kminor_lower, minor_limits_gpt_lower, minor_scales_with_density_lower,
            scale_by_complement_lower, idx_minor_lower, idx_minor_scaling_lower, kminor_start_lower,
            play, tlay, col_gas, fminor, jeta, jtemp, tropo, tau, (Float*)nullptr);

    idx_tropo = 0;
    host_launch(dim3(1, nlay, ncol), ref_gas::gas_optical_depths_minor_kernel<1,1,1>,
            ncol, nlay, ngpt, ngas, nflav, ntemp, neta, nminorupper, nminorkupper, idx_h2o, idx_tropo,
            gpoint_flavor, kminor_upper, minor_limits_gpt_upper, minor_scales_with_density_upper,
            scale_by_complement_upper, idx_minor_upper, idx_minor_scaling_upper, kminor_start_upper,
            play, tlay, col_gas, fminor, jeta, jtemp, tropo, tau, (Float*)nullptr);
}


void ref_compute_tau_rayleigh(
        const int ncol, const int nlay, const int nbnd, const int ngpt,
        const int ngas, const int nflav, const int neta, const int npres, const int ntemp,
        const int* gpoint_flavor, const int* band_lims_gpt, const Float* krayl,
        int idx_h2o, const Float* col_dry, const Float* col_gas,
        const Float* fminor, const int* jeta, const Bool* tropo, const int* jtemp,
        Float* tau_rayleigh)
{
    host_launch(dim3(ncol, nlay), ref_gas::compute_tau_rayleigh_kernel,
            ncol, nlay, nbnd, ngpt, ngas, nflav, neta, npres, ntemp,
            gpoint_flavor, band_lims_gpt, krayl, idx_h2o, col_dry, col_gas,
            fminor, jeta, tropo, jtemp, tau_rayleigh);
}


void ref_combine_abs_and_rayleigh(
        const int ncol, const int nlay, const int ngpt,
        const Float* tau_abs, const Float* tau_rayleigh,
        Float* tau, Float* ssa, Float* g)
{
    const Float tmin = std::numeric_limits<Float>::min();
    host_launch(dim3(ncol, nlay, ngpt), ref_gas::combine_abs_and_rayleigh_kernel,
            ncol, nlay, ngpt, tmin, tau_abs, tau_rayleigh, tau, ssa, g);
}


void ref_compute_planck_source(
        const int ncol, const int nlay, const int nbnd, const int ngpt,
        const int nflav, const int neta, const int npres, const int ntemp, const int nPlanckTemp,
        const Float* tlay, const Float* tlev, const Float* tsfc, const int sfc_lay,
        const Float* fmajor, const int* jeta, const Bool* tropo, const int* jtemp, const int* jpress,
        const int* gpoint_bands, const int* band_lims_gpt, const Float* pfracin,
        const Float temp_ref_min, const Float totplnk_delta, const Float* totplnk,
        const int* gpoint_flavor,
        Float* sfc_src, Float* lay_src, Float* lev_src, Float* sfc_src_jac)
{
    const Float delta_Tsurf = Float(1.);
    host_launch(dim3(ncol, nlay, ngpt), ref_gas::Planck_source_kernel,
            ncol, nlay, nbnd, ngpt, nflav, neta, npres, ntemp, nPlanckTemp,
            tlay, tlev, tsfc, sfc_lay, fmajor, jeta, tropo, jtemp, jpress,
            gpoint_bands, band_lims_gpt, pfracin, temp_ref_min, totplnk_delta, totplnk,
            gpoint_flavor, delta_Tsurf, sfc_src, lay_src, lev_src, sfc_src_jac);
}


void ref_increment_1scalar_by_1scalar(int ncol, int nlay, int ngpt, Float* tau_inout, const Float* tau_in)
{
    host_launch(dim3(ncol, nlay, ngpt), ref_opt::increment_1scalar_by_1scalar_kernel,
            ncol, nlay, ngpt, tau_inout, tau_in);
}

void ref_increment_2stream_by_2stream(
        int ncol, int nlay, int ngpt,
        Float* tau_inout, Float* ssa_inout, Float* g_inout,
        const Float* tau_in, const Float* ssa_in, const Float* g_in)
{
    const Float eps = std::numeric_limits<Float>::min() * Float(3.);
    host_launch(dim3(ncol, nlay, ngpt), ref_opt::increment_2stream_by_2stream_kernel,
            ncol, nlay, ngpt, eps, tau_inout, ssa_inout, g_inout, tau_in, ssa_in, g_in);
}

void ref_inc_1scalar_by_1scalar_bybnd(
        int ncol, int nlay, int ngpt, Float* tau_inout, const Float* tau_in,
        int nbnd, const int* band_lims_gpoint)
{
    host_launch(dim3(ncol, nlay, ngpt), ref_opt::inc_1scalar_by_1scalar_bybnd_kernel,
            ncol, nlay, ngpt, tau_inout, tau_in, nbnd, band_lims_gpoint);
}

void ref_inc_2stream_by_2stream_bybnd(
        int ncol, int nlay, int ngpt,
        Float* tau_inout, Float* ssa_inout, Float* g_inout,
        const Float* tau_in, const Float* ssa_in, const Float* g_in,
        int nbnd, const int* band_lims_gpoint)
{
    const Float eps = std::numeric_limits<Float>::min() * Float(3.);
    host_launch(dim3(ncol, nlay, ngpt), ref_opt::inc_2stream_by_2stream_bybnd_kernel,
            ncol, nlay, ngpt, eps, tau_inout, ssa_inout, g_inout, tau_in, ssa_in, g_in,
            nbnd, band_lims_gpoint);
}

void ref_delta_scale_2str_k(int ncol, int nlay, int ngpt, Float* tau_inout, Float* ssa_inout, Float* g_inout)
{
    const Float eps = std::numeric_limits<Float>::min() * Float(3.);
    host_launch(dim3(ncol, nlay, ngpt), ref_opt::delta_scale_2str_k_kernel,
            ncol, nlay, ngpt, eps, tau_inout, ssa_inout, g_inout);
}

void ref_sum_broadband(int ncol, int nlev, int ngpt, const Float* gpt_flux, Float* flux)
{
    host_launch(dim3(ncol, nlev), ref_flx::sum_broadband_kernel, ncol, nlev, ngpt, gpt_flux, flux);
}

void ref_net_broadband_precalc(int ncol, int nlev, const Float* flux_dn, const Float* flux_up, Float* flux_net)
{
    host_launch(dim3(ncol, nlev), ref_flx::net_broadband_precalc_kernel, ncol, nlev, flux_dn, flux_up, flux_net);
}

// boundary-condition and transpose launchers on their own (rte_solver_kernels_launchers.cu:20-45,
// gas_optics_rrtmgp_kernels_launchers.cu:20-58): kernels rte_solver_kernels.cu:351-387, gas_optics_rrtmgp_kernels.cu:76-111
void ref_apply_BC_0(int ncol, int nlay, int ngpt, Bool top_at_1, Float* gpt_flux_dn)
{
    host_launch(dim3(ncol, ngpt), static_cast<void(*)(const int, const int, const int, const Bool, Float*)>(ref_rte::apply_BC_kernel),
            ncol, nlay, ngpt, top_at_1, gpt_flux_dn);
}

void ref_apply_BC_gpt(int ncol, int nlay, int ngpt, Bool top_at_1, const Float* inc_flux, Float* gpt_flux_dn)
{
    host_launch(dim3(ncol, ngpt), static_cast<void(*)(const int, const int, const int, const Bool, const Float*, Float*)>(ref_rte::apply_BC_kernel),
            ncol, nlay, ngpt, top_at_1, inc_flux, gpt_flux_dn);
}

void ref_apply_BC_factor(int ncol, int nlay, int ngpt, Bool top_at_1, const Float* inc_flux, const Float* factor, Float* gpt_flux_dn)
{
    host_launch(dim3(ncol, ngpt), static_cast<void(*)(const int, const int, const int, const Bool, const Float*, const Float*, Float*)>(ref_rte::apply_BC_kernel),
            ncol, nlay, ngpt, top_at_1, inc_flux, factor, gpt_flux_dn);
}

void ref_reorder123x321(int ni, int nj, int nk, const Float* arr_in, Float* arr_out)
{
    host_launch(dim3(ni, nj, nk), ref_gas::reorder123x321_kernel, ni, nj, nk, arr_in, arr_out);
}

void ref_reorder12x21(int ni, int nj, const Float* arr_in, Float* arr_out)
{
    host_launch(dim3(ni, nj), ref_gas::reorder12x21_kernel, ni, nj, arr_in, arr_out);
}
}
