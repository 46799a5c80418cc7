/* Rte_solver_kernels_cuda -- same namespace, names and argument lists as the reference's device-side boundary
 * (/root/reference/include_kernels_cuda/rte_solver_kernels_cuda.h:33-64), forwarding to librrx_hip.so. */
#ifndef RTE_SOLVER_KERNELS_CUDA_H
#define RTE_SOLVER_KERNELS_CUDA_H
#include "rrx_forward.h"

namespace Rte_solver_kernels_cuda
{
    inline void apply_BC(const int ncol, const int nlay, const int ngpt, const Bool top_at_1,
                  const Float* inc_flux_dir, const Float* mu0, Float* gpt_flux_dir)
    { RRX_CALL(rrx_apply_BC_factor, ncol, nlay, ngpt, top_at_1, inc_flux_dir, mu0, gpt_flux_dir); }

    inline void apply_BC(const int ncol, const int nlay, const int ngpt, const Bool top_at_1, Float* gpt_flux_dn)
    { RRX_CALL(rrx_apply_BC_0, ncol, nlay, ngpt, top_at_1, gpt_flux_dn); }

    inline void apply_BC(const int ncol, const int nlay, const int ngpt, const Bool top_at_1, const Float* inc_flux_dif, Float* gpt_flux_dn)
    { RRX_CALL(rrx_apply_BC_gpt, ncol, nlay, ngpt, top_at_1, inc_flux_dif, gpt_flux_dn); }

    inline void sw_solver_2stream(
            const int ncol, const int nlay, const int ngpt, const Bool top_at_1,
            const Float* tau, const Float* ssa, const Float* g,
            const Float* mu0,
            const Float* sfc_alb_dir, const Float* sfc_alb_dif,
            const Float* inc_flux_dir,
            Float* flux_up, Float* flux_dn, Float* flux_dir,
            const Bool has_dif_bc, const Float* inc_flux_dif,
            const Bool do_broadband, Float* flux_up_loc, Float* flux_dn_loc, Float* flux_dir_loc)
    {
        RRX_CALL(rrx_sw_solver_2stream, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif, inc_flux_dir,
                 flux_up, flux_dn, flux_dir, has_dif_bc, inc_flux_dif, do_broadband, flux_up_loc, flux_dn_loc, flux_dir_loc);
    }

    inline void lw_solver_noscat(
            const int ncol, const int nlay, const int ngpt, const Bool top_at_1, const int nmus,
            const Float* secants, const Float* weights,
            const Float* tau, const Float* lay_source,
            const Float* lev_source,
            const Float* sfc_emis, const Float* sfc_src,
            const Float* inc_flux,
            Float* flux_up, Float* flux_dn,
            const Bool do_broadband, Float* flux_up_loc, Float* flux_dn_loc,
            const Bool do_jacobians, const Float* sfc_src_jac, Float* flux_up_jac)
    {
        RRX_CALL(rrx_lw_solver_noscat, ncol, nlay, ngpt, top_at_1, nmus, secants, weights, tau, lay_source, lev_source,
                 sfc_emis, sfc_src, inc_flux, flux_up, flux_dn, do_broadband, flux_up_loc, flux_dn_loc,
                 do_jacobians, sfc_src_jac, flux_up_jac);
    }

    inline void lw_secants_array(
            const int ncol, const int ngpt, const int n_quad_angs, const int max_gauss_pts,
            const Float* Gauss_Ds, Float* secants)
    { RRX_CALL(rrx_lw_secants_array, ncol, ngpt, n_quad_angs, max_gauss_pts, Gauss_Ds, secants); }
}
#endif
