/* C entry points around Radiation_solver_longwave / _shortwave (the reference's class structure: include_test/Radiation_solver.h =
 * /root/reference/include_test/Radiation_solver.h:33-235) for a host written in another language: librte_rrtmgp_hip.so (fp64) /
 * librte_rrtmgp_hip_sp.so (fp32), implementation rte-rrtmgp-cpp_amd/host/src_test/cxx_driver_api.cpp. `Real` is double resp. float.
 * Every function returns 0 on success (create: a handle or NULL); rrx_cxx_driver_error() holds the message of the calling thread.
 * Nothing here synchronises the device. Binding example: INTEGRATION.md section 4, rte-rrtmgp-cpp_amd/cxx_driver.py. */
#ifndef RRX_CXX_DRIVER_H
#define RRX_CXX_DRIVER_H
#ifdef RTE_USE_SP
typedef float Real;
#else
typedef double Real;
#endif
#ifdef __cplusplus
extern "C" {
#endif
const char* rrx_cxx_driver_error(void);
/* loads coefficients_lw.nc, coefficients_sw.nc [, cloud_coefficients_{lw,sw}.nc] from `dir` (the file names of the reference's driver);
   gas_names: the gases the caller will provide; top_at_1: 1 = columns ordered from the top down (stated, so that no solve reads back) */
void* rrx_cxx_driver_create(const char* dir, int ngas, const char* const* gas_names, int clouds, int top_at_1);
void  rrx_cxx_driver_destroy(void* handle);
/* vmr: DEVICE pointer to an (n1, n2) array, column index fastest: (1,1) scalar, (1,nlay) profile or (ncol,nlay) field; copied */
int   rrx_cxx_driver_set_gas(void* handle, const char* name, const Real* vmr, int n1, int n2);
/* set_column_block / set_broadband_solvers / set_column_sorting (-1 auto, 0, 1) / set_column_padding of both solvers */
int   rrx_cxx_driver_settings(void* handle, int column_block, int broadband, int sort_mode, int pad);
/* one LW + one SW solve_gpu (fluxes only) enqueued on `stream`; DEVICE arrays: (ncol,nlay) / (ncol,nlay+1) fields, (ncol) vectors,
   surface properties (nbnd,ncol); lwp, iwp, rel, dei NULL without clouds; out7: seven (ncol, nlay+1) arrays for LW up, dn, net and
   SW up, dn, dn_dir, net, or NULL (the driver then keeps them: rrx_cxx_driver_fluxes) */
int   rrx_cxx_driver_solve(void* handle, int ncol, int nlay, int nbnd_lw, int nbnd_sw,
        const Real* p_lay, const Real* p_lev, const Real* t_lay, const Real* t_lev, const Real* t_sfc,
        const Real* emis_sfc, const Real* sfc_alb_dir, const Real* sfc_alb_dif, const Real* tsi_scaling, const Real* mu0,
        const Real* lwp, const Real* iwp, const Real* rel, const Real* dei, Real* const* out7, void* stream);
int   rrx_cxx_driver_fluxes(void* handle, const Real** ptrs7);
#ifdef __cplusplus
}
#endif
#endif
