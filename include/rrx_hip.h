/*
 * rrx_hip.h -- C ABI of librrx_hip.so, the MI355X (gfx950) device layer of the RTE+RRTMGP hot path.
 *
 * This is the drop-in boundary: every entry point replaces ONE launcher of the
 * reference's device-side boundary `include_kernels_cuda/{...}.h` (30 functions in 5 namespaces) or ONE of the small
 * inline kernels of `src_cuda/{...}.cu`. The reference's namespaces (Rte_solver_kernels_cuda, ...) are kept as
 * header-only forwarders onto these symbols in include/rte_solver_kernels_cuda.h etc., so driver code written
 * against the reference compiles unchanged.
 *
 * Conventions (identical to the reference launchers unless stated):
 *   - plain C: scalars by value, arrays as raw DEVICE pointers, caller owns every array;
 *   - arrays are column-major with the column index fastest: (icol,ilay,igpt) at icol + ilay*ncol + igpt*ncol*nlay;
 *   - index-valued arrays hold 1-based values (band_lims_gpt, gpoint_flavor, jeta, jtemp, ...);
 *   - Bool is `signed char` (RTE_USE_CBOOL, the only setting any shipped reference config uses);
 *   - two precisions in one library: suffix _f64 (Float = double, reference default) and _f32 (RTE_USE_SP);
 *   - every function returns 0 on success, non-zero on error (message: rrx_last_error()); the C++ forwarders
 *     turn that into std::runtime_error, mirroring the reference's exception behaviour;
 *   - last argument `stream` is a hipStream_t passed as void* (NULL = the default stream, what the reference uses).
 *   - no function synchronises the device or allocates with hipMalloc; scratch (only rrx_*_solver in
 *     do_broadband mode) comes from the stream-ordered pool (hipMallocAsync).
 *
 * Citations are file:line under /root/reference.
 */
#ifndef RRX_HIP_H
#define RRX_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef signed char RrxBool;

/* ------------------------------------------------------------------ runtime ---------------------------------- */
/* replaces include/tools_gpu.h:72-99 (Tools_gpu::allocate_gpu/free_gpu), src_cuda/tools_gpu.cu, mem_pool_gpu.cu */
const char* rrx_last_error(void);
int rrx_device_count(int* n);
int rrx_set_device(int dev);
int rrx_malloc(void** ptr, unsigned long long bytes);
int rrx_free(void* ptr);
/* stream-ordered twins (what Array_gpu uses): allocation from the device's default memory pool with the release threshold lifted,
   so freed blocks are reused by the next solve and no call synchronises the device; the copies are enqueued on `stream` and awaited */
int rrx_malloc_async(void** ptr, unsigned long long bytes, void* stream);
int rrx_free_async(void* ptr, void* stream);
/* release of a block allocated under `alloc_stream` that may have been used under `release_stream` as well: the release stream waits
   for the allocation stream's work first; falls back to hipFree if the stream-ordered free fails (never leaks). The pool keeps
   freed blocks for reuse; RRX_POOL_RELEASE_THRESHOLD=<bytes> in the environment caps that (default: keep everything). */
int rrx_free_async_ordered(void* ptr, void* alloc_stream, void* release_stream);
int rrx_memcpy_h2d_stream(void* dst, const void* src, unsigned long long bytes, void* stream);
int rrx_memcpy_d2h_stream(void* dst, const void* src, unsigned long long bytes, void* stream);
int rrx_memcpy_h2d(void* dst, const void* src, unsigned long long bytes);
int rrx_memcpy_d2h(void* dst, const void* src, unsigned long long bytes);
int rrx_memcpy_d2d(void* dst, const void* src, unsigned long long bytes, void* stream);
int rrx_memset(void* dst, int value, unsigned long long bytes, void* stream);
int rrx_synchronize(void* stream);
int rrx_stream_create(void** stream);
int rrx_stream_destroy(void* stream);
/* The any-nlay solver forms (LW with several quadrature angles or Jacobians, columns of 288 layers and more, do_broadband outside
   the fused tilings) keep ONE grow-only block of device memory per (calling thread, device, stream) for their per-g-point
   temporaries. The stream owns it: rrx_stream_destroy returns it to the pool, and so does rrx_release_workspace (for streams the
   caller created itself -- call it from the thread that made the solver calls, before destroying the stream). A block larger than
   RRX_WORKSPACE_KEEP bytes (environment, default 32 GiB) is returned at the end of the call that used it. */
int rrx_release_workspace(void* stream);
/* perm[i] = min(i, ncol-1), i < ncol + npad: the identity order, padded (see rrx_sort_columns) */
int rrx_identity_columns(int ncol, int npad, int* perm, void* stream);
unsigned long long rrx_workspace_bytes(void* stream);
/* include/Array.h:311-350,579-622 (Array_gpu::subset / subset_kernel): N-D block gather, singleton dimensions are
   broadcast. sub_dims/strides/starts/spread are HOST arrays of length ndim (<= 7); strides in elements, starts 0-based. */
int rrx_subset_nd(void* out, const void* in, int elem_bytes, int ndim, const int* sub_dims, const long long* strides,
                  const int* starts, const int* spread, void* stream);
/* kernel-variant switches used by bench.py A/B runs (0 = default) */
int rrx_set_lw_variant(int v);
int rrx_set_sw_variant(int v);
/* column groups (8 or 16 columns x all levels) below which do_broadband falls back from the fused one-kernel form to
   per-g-point fluxes in a workspace + sum (default 512: measured break-even at C4 shapes is 256-512; 1 = always fused) */
int rrx_set_broadband_min_groups(int n);
/* g-point ranges per column group in the one-kernel broadband solvers: 0 (default) = as many (a power of two, at most 16) as it
   takes to reach the workgroup count above when columns are few, 1 = never split, n = n ranges. Partial sums are added in
   range order by a second kernel: deterministic, but not the association of the unsplit sum. */
int rrx_set_broadband_gsplit(int n);
/* 1 (default): the "direct" gas optics run the windowed kernel (LUT boxes staged in LDS) ahead of the gather kernel;
   0: gather kernel only (A/B runs, tests). Like the other switches it acts on the calling host thread. */
int rrx_set_gas_window(int on);
/* diagnostic: with RRX_GW_STATS set in the environment every windowed gas-optics launch waits for its kernel and counts the
   workgroups it handed back to the gather kernel; this returns (and optionally resets) the calling thread's totals */
int rrx_gas_window_stats(long long* handed_back, long long* workgroups, int reset);

#define RRX_DECLARE(F, SFX) \
/* ---- Rte_solver_kernels_cuda : include_kernels_cuda/rte_solver_kernels_cuda.h:33-64 ---- */ \
/* apply_BC x3: src_kernels_cuda/rte_solver_kernels_launchers.cu:20-45 */ \
int rrx_apply_BC_factor##SFX(int ncol, int nlay, int ngpt, RrxBool top_at_1, const F* inc_flux_dir, const F* mu0, F* gpt_flux_dir, void* stream); \
int rrx_apply_BC_0##SFX(int ncol, int nlay, int ngpt, RrxBool top_at_1, F* gpt_flux_dn, void* stream); \
int rrx_apply_BC_gpt##SFX(int ncol, int nlay, int ngpt, RrxBool top_at_1, const F* inc_flux_dif, F* gpt_flux_dn, void* stream); \
/* lw_secants_array: launchers.cu:48-58 */ \
int rrx_lw_secants_array##SFX(int ncol, int ngpt, int n_gauss_quad, int max_gauss_pts, const F* gauss_Ds, F* secants, void* stream); \
/* lw_solver_noscat: launchers.cu:61-286. Beyond the reference GPU path it honours do_broadband (flux_*_loc = \
   (ncol,nlay+1) g-point sums, the CPU/Fortran behaviour, src/Rte_lw.cpp:176), do_jacobians and nmus 1..4. */ \
int rrx_lw_solver_noscat##SFX( \
        int ncol, int nlay, int ngpt, RrxBool top_at_1, int nmus, \
        const F* secants, const F* weights, \
        const F* tau, const F* lay_source, const F* lev_source, \
        const F* sfc_emis, const F* sfc_src, const F* inc_flux, \
        F* flux_up, F* flux_dn, \
        RrxBool do_broadband, F* flux_up_loc, F* flux_dn_loc, \
        RrxBool do_jacobians, const F* sfc_src_jac, F* flux_up_jac, void* stream); \
/* sw_solver_2stream: launchers.cu:289-447. mu0 is (ncol) as on the reference GPU path; sfc_alb_dir is indexed \
   per g-point (Fortran semantics); has_dif_bc and do_broadband are honoured. g may be NULL = asymmetry \
   identically zero (what clear-sky gas optics produce): same fluxes as with an array of zeros, which is not read. */ \
int rrx_sw_solver_2stream##SFX( \
        int ncol, int nlay, int ngpt, RrxBool top_at_1, \
        const F* tau, const F* ssa, const F* g, const F* mu0, \
        const F* sfc_alb_dir, const F* sfc_alb_dif, const F* inc_flux_dir, \
        F* flux_up, F* flux_dn, F* flux_dir, \
        RrxBool has_dif_bc, const F* inc_flux_dif, \
        RrxBool do_broadband, F* flux_up_loc, F* flux_dn_loc, F* flux_dir_loc, void* stream); \
/* ---- Gas_optics_rrtmgp_kernels_cuda : include_kernels_cuda/gas_optics_rrtmgp_kernels_cuda.h:33-132 ---- */ \
int rrx_reorder123x321##SFX(int ni, int nj, int nk, const F* arr_in, F* arr_out, void* stream); \
int rrx_reorder12x21##SFX(int ni, int nj, const F* arr_in, F* arr_out, void* stream); \
int rrx_zero_array##SFX(int ni, int nj, int nk, F* arr, void* stream); \
/* interpolation: gas_optics_rrtmgp_kernels_launchers.cu:91-125 */ \
int rrx_interpolation##SFX( \
        int ncol, int nlay, int ngas, int nflav, int neta, int npres, int ntemp, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, \
        const F* vmr_ref, const F* play, const F* tlay, F* col_gas, \
        int* jtemp, F* fmajor, F* fminor, F* col_mix, RrxBool* tropo, int* jeta, int* jpress, void* stream); \
/* combine_abs_and_rayleigh: launchers.cu:128-165 (ssa threshold 2*epsilon: CPU semantics, src/Gas_optics_rrtmgp.cpp:378) */ \
int rrx_combine_abs_and_rayleigh##SFX(int ncol, int nlay, int ngpt, const F* tau_abs, const F* tau_rayleigh, F* tau, F* ssa, F* g, void* stream); \
/* compute_tau_rayleigh: launchers.cu:168-221 */ \
int rrx_compute_tau_rayleigh##SFX( \
        int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, \
        const int* gpoint_flavor, const int* band_lims_gpt, const F* krayl, \
        int idx_h2o, const F* col_dry, const F* col_gas, \
        const F* fminor, const int* jeta, const RrxBool* tropo, const int* jtemp, F* tau_rayleigh, void* stream); \
/* compute_tau_absorption: launchers.cu:234-438 (major + minor lower + minor upper, ADDED onto tau) */ \
int rrx_compute_tau_absorption##SFX( \
        int ncol, int nlay, int nband, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, \
        int nminorlower, int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, \
        const int* gpoint_flavor, const int* band_lims_gpt, \
        const F* kmajor, const F* kminor_lower, const F* kminor_upper, \
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper, \
        const RrxBool* minor_scales_with_density_lower, const RrxBool* minor_scales_with_density_upper, \
        const RrxBool* scale_by_complement_lower, const RrxBool* scale_by_complement_upper, \
        const int* idx_minor_lower, const int* idx_minor_upper, \
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper, \
        const int* kminor_start_lower, const int* kminor_start_upper, \
        const RrxBool* tropo, const F* col_mix, const F* fmajor, const F* fminor, \
        const F* play, const F* tlay, const F* col_gas, \
        const int* jeta, const int* jtemp, const int* jpress, F* tau, void* stream); \
/* addition: the same sum STORED into tau (what Gas_optics_rrtmgp_gpu needs after its zero fill: src_cuda/Gas_optics_rrtmgp.cu \
   compute_gas_taus; saves the fill and the read-back) */ \
int rrx_compute_tau_absorption_set##SFX( \
        int ncol, int nlay, int nband, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, \
        int nminorlower, int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, \
        const int* gpoint_flavor, const int* band_lims_gpt, \
        const F* kmajor, const F* kminor_lower, const F* kminor_upper, \
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper, \
        const RrxBool* minor_scales_with_density_lower, const RrxBool* minor_scales_with_density_upper, \
        const RrxBool* scale_by_complement_lower, const RrxBool* scale_by_complement_upper, \
        const int* idx_minor_lower, const int* idx_minor_upper, \
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper, \
        const int* kminor_start_lower, const int* kminor_start_upper, \
        const RrxBool* tropo, const F* col_mix, const F* fmajor, const F* fminor, \
        const F* play, const F* tlay, const F* col_gas, \
        const int* jeta, const int* jtemp, const int* jpress, F* tau, void* stream); \
/* compute_planck_source: launchers.cu:441-521 */ \
int rrx_compute_planck_source##SFX( \
        int ncol, int nlay, int nbnd, int ngpt, int nflav, int neta, int npres, int ntemp, int nPlanckTemp, \
        const F* tlay, const F* tlev, const F* tsfc, int sfc_lay, \
        const F* fmajor, const int* jeta, const RrxBool* tropo, const int* jtemp, const int* jpress, \
        const int* gpoint_bands, const int* band_lims_gpt, const F* pfracin, \
        F temp_ref_min, F totplnk_delta, const F* totplnk, const int* gpoint_flavor, \
        F* sfc_src, F* lay_src, F* lev_src, F* sfc_src_jac, void* stream); \
/* fused SW gas optics used by Gas_optics_rrtmgp_gpu (tau_abs + tau_rayleigh + combine in one pass, tau/ssa/g \
   written once; same arithmetic as the three launchers above called in sequence on a zeroed tau). g may be NULL: \
   the asymmetry parameter of the gas optics is identically zero and is then not written */ \
int rrx_gas_optics_sw_fused##SFX( \
        int ncol, int nlay, int nband, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, \
        int nminorlower, int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, \
        const int* gpoint_flavor, const int* band_lims_gpt, \
        const F* kmajor, const F* kminor_lower, const F* kminor_upper, \
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper, \
        const RrxBool* minor_scales_with_density_lower, const RrxBool* minor_scales_with_density_upper, \
        const RrxBool* scale_by_complement_lower, const RrxBool* scale_by_complement_upper, \
        const int* idx_minor_lower, const int* idx_minor_upper, \
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper, \
        const int* kminor_start_lower, const int* kminor_start_upper, \
        const RrxBool* tropo, const F* col_mix, const F* fmajor, const F* fminor, \
        const F* play, const F* tlay, const F* col_gas, const F* col_dry, \
        const int* jeta, const int* jtemp, const int* jpress, const F* krayl, \
        F* tau, F* ssa, F* g, void* stream); \
/* "direct" gas optics: the interpolation state (interpolation_kernel, gas_optics_rrtmgp_kernels.cu:317-395) is computed \
   inside the consumer from (play, tlay, col_gas) instead of being written by rrx_interpolation and read back by \
   rrx_compute_tau_absorption / rrx_compute_planck_source: same expressions in the same order, so the same bits, without \
   the seven intermediate arrays (jtemp, jpress, tropo, jeta, col_mix, fminor, fmajor). What Gas_optics_rrtmgp_gpu::gas_optics \
   (src_cuda/Gas_optics_rrtmgp.cu:907-1201) runs here. flavor(2,nflav), vmr_ref(2,0:ngas,ntemp) as for rrx_interpolation. */ \
int rrx_gas_optics_lw_direct##SFX( \
        int ncol, int nlay, int nband, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, \
        int nminorlower, int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, \
        const int* gpoint_flavor, const int* band_lims_gpt, \
        const F* kmajor, const F* kminor_lower, const F* kminor_upper, \
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper, \
        const RrxBool* minor_scales_with_density_lower, const RrxBool* minor_scales_with_density_upper, \
        const RrxBool* scale_by_complement_lower, const RrxBool* scale_by_complement_upper, \
        const int* idx_minor_lower, const int* idx_minor_upper, \
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper, \
        const int* kminor_start_lower, const int* kminor_start_upper, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, const F* vmr_ref, \
        const F* play, const F* tlay, const F* col_gas, F* tau, void* stream); \
/* rrx_gas_optics_lw_direct with optical properties given by band (clouds, aerosols: cld_* are (ncol,nlay,nbnd) arrays, bands delimited by \
   band_lims_gpt) added where the gas optics is stored -- increment_1scalar_by_1scalar_bybnd folded into the producer, same \
   arithmetic and bits, without reading and re-writing the g-point arrays; cld_tau = NULL: the plain entry */ \
int rrx_gas_optics_lw_direct_allsky##SFX( \
        int ncol, int nlay, int nband, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, \
        int nminorlower, int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, \
        const int* gpoint_flavor, const int* band_lims_gpt, \
        const F* kmajor, const F* kminor_lower, const F* kminor_upper, \
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper, \
        const RrxBool* minor_scales_with_density_lower, const RrxBool* minor_scales_with_density_upper, \
        const RrxBool* scale_by_complement_lower, const RrxBool* scale_by_complement_upper, \
        const int* idx_minor_lower, const int* idx_minor_upper, \
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper, \
        const int* kminor_start_lower, const int* kminor_start_upper, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, const F* vmr_ref, \
        const F* play, const F* tlay, const F* col_gas, F* tau, const F* cld_tau, void* stream); \
int rrx_gas_optics_sw_direct##SFX( \
        int ncol, int nlay, int nband, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, \
        int nminorlower, int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, \
        const int* gpoint_flavor, const int* band_lims_gpt, \
        const F* kmajor, const F* kminor_lower, const F* kminor_upper, \
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper, \
        const RrxBool* minor_scales_with_density_lower, const RrxBool* minor_scales_with_density_upper, \
        const RrxBool* scale_by_complement_lower, const RrxBool* scale_by_complement_upper, \
        const int* idx_minor_lower, const int* idx_minor_upper, \
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper, \
        const int* kminor_start_lower, const int* kminor_start_upper, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, const F* vmr_ref, \
        const F* play, const F* tlay, const F* col_gas, const F* col_dry, const F* krayl, \
        F* tau, F* ssa, F* g, void* stream); \
/* rrx_gas_optics_sw_direct with optical properties given by band (clouds, aerosols: cld_* are (ncol,nlay,nbnd) arrays, bands delimited by \
   band_lims_gpt) added where the gas optics is stored -- increment_2stream_by_2stream_bybnd folded into the producer, same \
   arithmetic and bits, without reading and re-writing the g-point arrays (g must be given); cld_tau = NULL: the plain entry */ \
int rrx_gas_optics_sw_direct_allsky##SFX( \
        int ncol, int nlay, int nband, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, \
        int nminorlower, int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, \
        const int* gpoint_flavor, const int* band_lims_gpt, \
        const F* kmajor, const F* kminor_lower, const F* kminor_upper, \
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper, \
        const RrxBool* minor_scales_with_density_lower, const RrxBool* minor_scales_with_density_upper, \
        const RrxBool* scale_by_complement_lower, const RrxBool* scale_by_complement_upper, \
        const int* idx_minor_lower, const int* idx_minor_upper, \
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper, \
        const int* kminor_start_lower, const int* kminor_start_upper, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, const F* vmr_ref, \
        const F* play, const F* tlay, const F* col_gas, const F* col_dry, const F* krayl, \
        F* tau, F* ssa, F* g, const F* cld_tau, const F* cld_ssa, const F* cld_g, void* stream); \
int rrx_planck_source_direct##SFX( \
        int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, int nPlanckTemp, \
        const F* play, const F* tlay, const F* tlev, const F* tsfc, int sfc_lay, const F* col_gas, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, const F* vmr_ref, \
        const int* gpoint_bands, const int* band_lims_gpt, const F* pfracin, \
        F totplnk_delta, const F* totplnk, const int* gpoint_flavor, \
        F* sfc_src, F* lay_src, F* lev_src, F* sfc_src_jac, void* stream); \
/* "Planck-lite" LW chain (what Gas_optics_rrtmgp_gpu::source + Rte_lw_gpu::rte_lw run here in broadband mode): \
   rrx_planck_fractions writes the Planck fractions pfrac(ncol,nlay,ngpt), the band Planck functions B(tlay)(ncol,nlay,nbnd) and \
   B(tlev)(ncol,nlay+1,nbnd) and the surface terms -- Planck_source_kernel (gas_optics_rrtmgp_kernels.cu:196-314) without its two \
   products; rrx_lw_solver_noscat_fractions (one quadrature angle, broadband fluxes) forms lay_source = pfrac*B_lay and \
   lev_source = sqrt(pfrac*pfrac')*B_lev on the fly; rrx_planck_sources_from_fractions materialises them for any other consumer \
   (bit-identical to rrx_compute_planck_source). */ \
int rrx_planck_fractions##SFX( \
        int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, int nPlanckTemp, \
        const F* play, const F* tlay, const F* tlev, const F* tsfc, int sfc_lay, const F* col_gas, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, const F* vmr_ref, \
        const int* gpoint_bands, const int* band_lims_gpt, const F* pfracin, \
        F totplnk_delta, const F* totplnk, const int* gpoint_flavor, \
        F* pfrac, F* blay, F* blev, F* sfc_src, F* sfc_src_jac, void* stream); \
/* rrx_gas_optics_lw_direct + rrx_planck_fractions in one pass over the cells: the interpolation state and the LUT windows are \
   shared -- the LW gas optics of Gas_optics_rrtmgp_gpu in broadband mode */ \
int rrx_gas_optics_lw_fractions##SFX( \
        int ncol, int nlay, int nband, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, int nPlanckTemp, \
        int nminorlower, int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, \
        const int* gpoint_flavor, const int* band_lims_gpt, const int* gpoint_bands, \
        const F* kmajor, const F* kminor_lower, const F* kminor_upper, \
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper, \
        const RrxBool* minor_scales_with_density_lower, const RrxBool* minor_scales_with_density_upper, \
        const RrxBool* scale_by_complement_lower, const RrxBool* scale_by_complement_upper, \
        const int* idx_minor_lower, const int* idx_minor_upper, \
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper, \
        const int* kminor_start_lower, const int* kminor_start_upper, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, const F* vmr_ref, \
        const F* play, const F* tlay, const F* tlev, const F* tsfc, int sfc_lay, const F* col_gas, \
        const F* pfracin, F totplnk_delta, const F* totplnk, \
        F* tau, F* pfrac, F* blay, F* blev, F* sfc_src, F* sfc_src_jac, void* stream); \
/* rrx_gas_optics_lw_fractions with optical properties given by band (clouds, aerosols: cld_* are (ncol,nlay,nbnd) arrays, bands delimited by \
   band_lims_gpt) added where the gas optics is stored -- increment_1scalar_by_1scalar_bybnd folded into the producer, same \
   arithmetic and bits, without reading and re-writing the g-point arrays; cld_tau = NULL: the plain entry */ \
int rrx_gas_optics_lw_fractions_allsky##SFX( \
        int ncol, int nlay, int nband, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, int nPlanckTemp, \
        int nminorlower, int nminorklower, int nminorupper, int nminorkupper, int idx_h2o, \
        const int* gpoint_flavor, const int* band_lims_gpt, const int* gpoint_bands, \
        const F* kmajor, const F* kminor_lower, const F* kminor_upper, \
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper, \
        const RrxBool* minor_scales_with_density_lower, const RrxBool* minor_scales_with_density_upper, \
        const RrxBool* scale_by_complement_lower, const RrxBool* scale_by_complement_upper, \
        const int* idx_minor_lower, const int* idx_minor_upper, \
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper, \
        const int* kminor_start_lower, const int* kminor_start_upper, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, const F* vmr_ref, \
        const F* play, const F* tlay, const F* tlev, const F* tsfc, int sfc_lay, const F* col_gas, \
        const F* pfracin, F totplnk_delta, const F* totplnk, \
        F* tau, F* pfrac, F* blay, F* blev, F* sfc_src, F* sfc_src_jac, const F* cld_tau, void* stream); \
int rrx_planck_sources_from_fractions##SFX(int ncol, int nlay, int ngpt, const int* gpoint_bands, const F* pfrac, const F* blay, \
        const F* blev, F* lay_src, F* lev_src, void* stream); \
int rrx_lw_solver_noscat_fractions##SFX( \
        int ncol, int nlay, int ngpt, RrxBool top_at_1, const F* secants, const F* weights, \
        const F* tau, const F* pfrac, const F* blay, const F* blev, const int* gpoint_bands, \
        const F* sfc_emis, const F* sfc_src, const F* inc_flux, F* flux_up_loc, F* flux_dn_loc, void* stream); \
/* ---- Optical_props_kernels_cuda : include_kernels_cuda/optical_props_kernels_cuda.h:33-56 ---- */ \
int rrx_increment_1scalar_by_1scalar##SFX(int ncol, int nlay, int ngpt, F* tau_inout, const F* tau_in, void* stream); \
int rrx_increment_2stream_by_2stream##SFX(int ncol, int nlay, int ngpt, F* tau_inout, F* ssa_inout, F* g_inout, const F* tau_in, const F* ssa_in, const F* g_in, void* stream); \
int rrx_inc_1scalar_by_1scalar_bybnd##SFX(int ncol, int nlay, int ngpt, F* tau_inout, const F* tau_in, int nbnd, const int* band_lims_gpoint, void* stream); \
int rrx_inc_2stream_by_2stream_bybnd##SFX(int ncol, int nlay, int ngpt, F* tau_inout, F* ssa_inout, F* g_inout, const F* tau_in, const F* ssa_in, const F* g_in, int nbnd, const int* band_lims_gpoint, void* stream); \
int rrx_delta_scale_2str_k##SFX(int ncol, int nlay, int ngpt, F* tau_inout, F* ssa_inout, F* g_inout, void* stream); \
/* ---- Fluxes_kernels_cuda : include_kernels_cuda/fluxes_kernels_cuda.h:33-51 ---- */ \
int rrx_sum_broadband##SFX(int ncol, int nlev, int ngpt, const F* gpt_flux, F* flux, void* stream); \
int rrx_net_broadband_precalc##SFX(int ncol, int nlev, const F* flux_dn, const F* flux_up, F* flux_net, void* stream); \
/* host-model coupling (no counterpart in the reference library): layer heating rate [K/s] from the net (down - up) \
   broadband flux flux_net(ncol,nlay+1) and the level pressures plev(ncol,nlay+1): -(g/cp) * dF_net/dp, either vertical ordering */ \
int rrx_heating_rate##SFX(int ncol, int nlay, F g_over_cp, const F* flux_net, const F* plev, F* heating_rate, void* stream); \
/* by-band: Fortran semantics (src_kernels/mo_fluxes_byband_kernels.F90:22-71; the CUDA text is buggy): \
   band_lims is (2,nbnd), 1-based inclusive, gpt_flux is the SPECTRAL (ncol,nlev,ngpt) array */ \
int rrx_sum_byband##SFX(int ncol, int nlev, int ngpt, int nbnd, const int* band_lims, const F* gpt_flux, F* bnd_flux, void* stream); \
int rrx_net_byband_full##SFX(int ncol, int nlev, int ngpt, int nbnd, const int* band_lims, const F* gpt_flux_dn, const F* gpt_flux_up, F* bnd_flux_net, void* stream); \
/* ---- Subset_kernels_cuda : include_kernels_cuda/subset_kernels_cuda.h:33-56 (n = 1..4 arrays at once) ---- */ \
int rrx_get_from_subset##SFX(int ncol, int nlay, int nbnd, int ncol_in, int col_s_in, int narr, \
        F* const* var_full, const F* const* var_sub, void* stream); \
/* ---- small kernels the reference keeps inside its host classes ---- */ \
/* src_cuda/Gas_optics_rrtmgp.cu:392-422 fill_gases_kernel (one gas per call, igas = 0 copies col_dry) */ \
int rrx_fill_gases##SFX(int ncol, int nlay, int dim1, int dim2, int ngas, int igas, F* vmr_out, const F* vmr_in, F* col_gas, const F* col_dry, void* stream); \
/* the same for all gases in one launch: vmr_in = HOST array of ngas device pointers, each (dim1[i], dim2[i]) = (1,1) scalar, \
   (1,nlay) profile or (ncol,nlay) field; col_gas(ncol,nlay,0:ngas) with slot 0 = col_dry (the per-gas vmr copy is not produced) */ \
int rrx_fill_gases_all##SFX(int ncol, int nlay, int ngas, const F* const* vmr_in, const int* dim1, const int* dim2, F* col_gas, const F* col_dry, void* stream); \
/* src_cuda/Gas_optics_rrtmgp.cu:806-903 get_col_dry (three kernels fused) */ \
int rrx_get_col_dry##SFX(int ncol, int nlay, const F* vmr_h2o, const F* plev, F* col_dry, void* stream); \
/* src_cuda/Rte_lw.cu:37-56, Rte_sw.cu:34-54 expand_and_transpose: (nbnd,ncol) -> (ncol,ngpt) */ \
int rrx_expand_and_transpose##SFX(int ncol, int nbnd, const int* band_lims_gpt, const F* arr_in, F* arr_out, void* stream); \
/* src_cuda/Gas_optics_rrtmgp.cu spread_col: toa_src(icol,igpt) = solar_source(igpt) */ \
int rrx_spread_col##SFX(int ncol, int ngpt, F* toa_src, const F* solar_source, void* stream); \
/* src_test/Radiation_solver.cu scaling_to_subset: toa_src(icol,igpt) *= tsi_scaling(icol) */ \
int rrx_scaling_to_subset##SFX(int ncol, int ngpt, F* toa_src, const F* tsi_scaling, void* stream); \
/* the two above in one pass: toa_src(icol,igpt) = solar_source(igpt) * tsi_scaling(icol) (tsi_scaling = NULL: no scaling) */ \
int rrx_toa_source##SFX(int ncol, int ngpt, F* toa_src, const F* solar_source, const F* tsi_scaling, void* stream); \
/* src_cuda/Aerosol_optics.cu:36-263 + Aerosol_optics_gpu::aerosol_optics (:305-345): CAMS aerosol optics per band in one kernel. \
   aermr: HOST array of 11 device pointers (aermr01..aermr11), each (ncol,nlay) or, where aermr_per_column[a] == 0, one (nlay) \
   profile shared by all columns (the reference materialises the broadcast, fill_aerosols_3d); aermr_per_column may be NULL \
   (all per column). rh, tau, ssa, g: (ncol,nlay[,nbnd]); plev (ncol,nlay+1); rh_upper (nhum); hydrophobic tables \
   (nbnd,nphobic), hydrophilic tables (nbnd,nhum,nphilic), band index fastest (Radiation_solver.cu:366-401) */ \
int rrx_aerosol_optics##SFX(int ncol, int nlay, int nbnd, int nhum, int nphobic, int nphilic, const F* const* aermr, const int* aermr_per_column, \
        const F* rh, const F* plev, const F* rh_upper, const F* mext_phobic, const F* ssa_phobic, const F* g_phobic, \
        const F* mext_philic, const F* ssa_philic, const F* g_philic, F* tau, F* ssa, F* g, void* stream); \
/* src_cuda/Cloud_optics.cu:31-127,181-329: LUT cloud optics per band; luts are (nsize,nbnd) */ \
int rrx_cloud_optics_2str##SFX(int ncol, int nlay, int nbnd, int nsize_liq, int nsize_ice, \
        F radliq_lwr, F radliq_upr, F diamice_lwr, F diamice_upr, \
        const F* lut_extliq, const F* lut_ssaliq, const F* lut_asyliq, \
        const F* lut_extice, const F* lut_ssaice, const F* lut_asyice, \
        const F* clwp, const F* ciwp, const F* reliq, const F* deice, F* tau, F* ssa, F* g, void* stream); \
/* rrx_cloud_optics_2str followed by rrx_delta_scale_2str_k (the reference driver's pair, Radiation_solver.cu:773-792) in one pass */ \
int rrx_cloud_optics_2str_delta##SFX(int ncol, int nlay, int nbnd, int nsize_liq, int nsize_ice, \
        F radliq_lwr, F radliq_upr, F diamice_lwr, F diamice_upr, \
        const F* lut_extliq, const F* lut_ssaliq, const F* lut_asyliq, \
        const F* lut_extice, const F* lut_ssaice, const F* lut_asyice, \
        const F* clwp, const F* ciwp, const F* reliq, const F* deice, F* tau, F* ssa, F* g, void* stream); \
int rrx_cloud_optics_1scl##SFX(int ncol, int nlay, int nbnd, int nsize_liq, int nsize_ice, \
        F radliq_lwr, F radliq_upr, F diamice_lwr, F diamice_upr, \
        const F* lut_extliq, const F* lut_ssaliq, const F* lut_asyliq, \
        const F* lut_extice, const F* lut_ssaice, const F* lut_asyice, \
        const F* clwp, const F* ciwp, const F* reliq, const F* deice, F* tau, void* stream); \
/* include/Array.h:311-350,579-622: column-range gather (subset) of an array whose FIRST dimension is the column: \
   out(icol, r) = in(col_s-1+icol, r), r < nrest */ \
int rrx_subset_cols##SFX(int ncol_full, int nrest, int col_s, int ncol_sub, const F* in, F* out, void* stream); \
/* same for arrays whose LAST dimension is the column, e.g. emis_sfc(nbnd,ncol) */ \
int rrx_subset_lastdim##SFX(int n1, int col_s, int ncol_sub, const F* in, F* out, void* stream); \
int rrx_fill##SFX(unsigned long long n, F value, F* arr, void* stream); \
/* ---- column ordering of the product chain (csrc/rrx_columns.hip; no counterpart in the reference library): columns are independent, so \
   a solve may process them in any order. perm (ncol + npad ints on the device) is a gather index: rrx_sort_columns = ascending order of \
   key(ncol) (the surface pressure: neighbouring columns then share LUT boxes in the windowed gas optics), its last npad entries repeat \
   the last column (padding to a multiple of 16 columns); rrx_column_spread sets flag = 1 where a run of `block` consecutive columns \
   spans more than threshold x its mean. gather: out(i, r) = in(perm[i], r), i < nout, column FIRST (fastest) dimension; gather_lastdim: \
   out(b, i) = in(b, perm[i]) for (n1, ncol) arrays; scatter: out(perm[i], r) = in(i, r), i < n (arrays of ncol_src / ncol_dst columns). */ \
int rrx_sort_columns##SFX(int ncol, const F* key, int npad, int* perm, void* stream); \
int rrx_column_spread##SFX(int ncol, const F* key, int block, F threshold, int* flag, void* stream); \
int rrx_gather_cols##SFX(int nout, unsigned long long nrest, const int* perm, int ncol_in, const F* in, F* out, void* stream); \
int rrx_scatter_cols##SFX(int n, unsigned long long nrest, const int* perm, int ncol_src, const F* in, int ncol_dst, F* out, void* stream); \
int rrx_gather_lastdim##SFX(int n1, int nout, const int* perm, const F* in, F* out, void* stream);

RRX_DECLARE(double, _f64)
RRX_DECLARE(float, _f32)

#ifdef __cplusplus
}
#endif
#endif
