// librrtmgp_kernels_hip.so -- the reference's CPU boundary (include/rrtmgp_kernels.h, the 19 bind(C) names of
// /root/reference/include/rrtmgp_kernels.h:32-289) on top of the MI355X device layer librrx_hip.so.
//
// Every entry point: host arrays -> device (stream-ordered pool, ONE stream per calling thread, created at the thread's first call
// and destroyed -- with the workspace block the any-nlay solver forms keep on it -- when the thread ends) -> the rrx_* launcher that
// replaces the Fortran kernel -> outputs back to the host -> wait. It is a compatibility surface for code written against
// the CPU API (the reference's own src/*.cpp link against it unchanged); the fast path keeps its data resident and uses
// include/rrx_hip.h or the _gpu classes directly. No CPU fallback: without a GPU every call throws.
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "rrtmgp_kernels.h"
#include "rrx_hip.h"

namespace
{
    [[noreturn]] void fail(const std::string& what) { throw std::runtime_error("rrtmgp_kernels_hip: " + what); }
    void ok(const int status) { if (status != 0) fail(rrx_last_error()); }

    // the calling thread's stream (round 4, ADVICE r03: a stream per CALL left the solvers' cached workspace block behind a dead
    // handle each time; rrx_stream_destroy now hands that block back, and the stream lives as long as its thread)
    struct ThreadStream
    {
        void* s = nullptr;
        ThreadStream() { ok(rrx_stream_create(&s)); }
        ~ThreadStream() { if (s != nullptr) rrx_stream_destroy(s); }
    };
    void* thread_stream() { static thread_local ThreadStream t; return t.s; }

    // Device staging of one call. Buffers go back to the pool in stream order when the call ends (also on an exception).
    class Stage
    {
        public:
            Stage() : stream_(thread_stream()) {}
            ~Stage()
            {
                for (void* p : bufs_) rrx_free_async(p, stream_);
                rrx_synchronize(stream_);
            }
            Stage(const Stage&) = delete;
            Stage& operator=(const Stage&) = delete;

            void* stream() const { return stream_; }

            template<typename T> T* alloc(const size_t n)
            {
                void* p = nullptr;
                ok(rrx_malloc_async(&p, n*sizeof(T), stream_));
                bufs_.push_back(p);
                return static_cast<T*>(p);
            }
            // upload; a null host pointer (an absent optional array) stays null
            template<typename T> T* in(const T* host, const size_t n)
            {
                if (host == nullptr) return nullptr;
                T* d = alloc<T>(n);
                ok(rrx_memcpy_h2d_stream(d, host, n*sizeof(T), stream_));
                return d;
            }
            // output only: allocate now, copy back in finish()
            template<typename T> T* out(T* host, const size_t n)
            {
                if (host == nullptr) return nullptr;
                T* d = alloc<T>(n);
                pending_.push_back({host, d, n*sizeof(T)});
                return d;
            }
            template<typename T> T* inout(T* host, const size_t n)
            {
                T* d = in<T>(host, n);
                if (d) pending_.push_back({host, d, n*sizeof(T)});
                return d;
            }
            void finish()
            {
                for (const Back& b : pending_) ok(rrx_memcpy_d2h_stream(b.host, b.dev, b.bytes, stream_));
                pending_.clear();
                ok(rrx_synchronize(stream_));
            }
        private:
            struct Back { void* host; const void* dev; size_t bytes; };
            void* stream_ = nullptr;
            std::vector<void*> bufs_;
            std::vector<Back> pending_;
    };

    inline size_t sz(const int a, const int b = 1, const int c = 1, const int d = 1)
    { return size_t(a)*size_t(b)*size_t(c)*size_t(d); }
}

#define RRX_K(name, ...) ok(RRX_SFX(name)(__VA_ARGS__, S.stream()))

namespace rrtmgp_kernels
{
extern "C"
{
// ---------------------------------------------------------------- fluxes (src/Fluxes.cpp:39-78)
void rte_sum_broadband(int* ncol, int* nlev, int* ngpt, Float* spectral_flux, Float* broadband_flux)
{
    Stage S;
    const Float* s = S.in(spectral_flux, sz(*ncol, *nlev, *ngpt));
    Float* b = S.out(broadband_flux, sz(*ncol, *nlev));
    RRX_K(rrx_sum_broadband, *ncol, *nlev, *ngpt, s, b);
    S.finish();
}

void rte_net_broadband_precalc(int* ncol, int* nlev, Float* flux_dn, Float* flux_up, Float* flux_net)
{
    Stage S;
    const size_t n = sz(*ncol, *nlev);
    const Float* dn = S.in(flux_dn, n); const Float* up = S.in(flux_up, n);
    Float* net = S.out(flux_net, n);
    RRX_K(rrx_net_broadband_precalc, *ncol, *nlev, dn, up, net);
    S.finish();
}

void sum_byband(int* ncol, int* nlev, int* ngpt, int* nbnd, int* band_lims, Float* spectral_flux, Float* byband_flux)
{
    Stage S;
    const int* lims = S.in(band_lims, sz(2, *nbnd));
    const Float* s = S.in(spectral_flux, sz(*ncol, *nlev, *ngpt));
    Float* b = S.out(byband_flux, sz(*ncol, *nlev, *nbnd));
    RRX_K(rrx_sum_byband, *ncol, *nlev, *ngpt, *nbnd, lims, s, b);
    S.finish();
}

void net_byband_precalc(int* ncol, int* nlev, int* nbnd, Float* byband_flux_dn, Float* byband_flux_up, Float* byband_flux_net)
{
    Stage S;                                    // element-wise dn - up over (ncol, nlev, nbnd)
    const size_t n = sz(*ncol, *nlev, *nbnd);
    const Float* dn = S.in(byband_flux_dn, n); const Float* up = S.in(byband_flux_up, n);
    Float* net = S.out(byband_flux_net, n);
    RRX_K(rrx_net_broadband_precalc, *ncol, (*nlev)*(*nbnd), dn, up, net);
    S.finish();
}

// host arrays: nothing to launch
void zero_array_3D(int* ni, int* nj, int* nk, Float* array) { std::memset(array, 0, sz(*ni, *nj, *nk)*sizeof(Float)); }
void zero_array_4D(int* ni, int* nj, int* nk, int* nl, Float* array) { std::memset(array, 0, sz(*ni, *nj, *nk, *nl)*sizeof(Float)); }

// ---------------------------------------------------------------- gas optics (src/Gas_optics_rrtmgp.cpp:906-1070)
void rrtmgp_interpolation(
        int* ncol, int* nlay, int* ngas, int* nflav, int* neta, int* npres, int* ntemp,
        int* flavor, Float* press_ref_log, Float* temp_ref,
        Float* press_ref_log_delta, Float* temp_ref_min, Float* temp_ref_delta, Float* press_ref_trop_log,
        Float* vmr_ref, Float* play, Float* tlay, Float* col_gas,
        int* jtemp, Float* fmajor, Float* fminor, Float* col_mix, Bool* tropo, int* jeta, int* jpress)
{
    Stage S;
    const size_t n = sz(*ncol, *nlay), nf = n*size_t(*nflav);
    const int* d_flavor = S.in(flavor, sz(2, *nflav));
    const Float* d_pref = S.in(press_ref_log, *npres); const Float* d_tref = S.in(temp_ref, *ntemp);
    const Float* d_vmr = S.in(vmr_ref, sz(2, *ngas + 1, *ntemp));
    const Float* d_play = S.in(play, n); const Float* d_tlay = S.in(tlay, n);
    Float* d_colgas = S.in(col_gas, n*size_t(*ngas + 1));
    int* d_jtemp = S.out(jtemp, n); Float* d_fmajor = S.out(fmajor, 8*nf); Float* d_fminor = S.out(fminor, 4*nf);
    Float* d_colmix = S.out(col_mix, 2*nf); Bool* d_tropo = S.out(tropo, n); int* d_jeta = S.out(jeta, 2*nf); int* d_jpress = S.out(jpress, n);
    RRX_K(rrx_interpolation, *ncol, *nlay, *ngas, *nflav, *neta, *npres, *ntemp, d_flavor, d_pref, d_tref,
          *press_ref_log_delta, *temp_ref_min, *temp_ref_delta, *press_ref_trop_log, d_vmr, d_play, d_tlay, d_colgas,
          d_jtemp, d_fmajor, d_fminor, d_colmix, d_tropo, d_jeta, d_jpress);
    S.finish();
}

void rrtmgp_compute_tau_absorption(
        int* ncol, int* nlay, int* nband, int* ngpt, int* ngas, int* nflav, int* neta, int* npres, int* ntemp,
        int* nminorlower, int* nminorklower, int* nminorupper, int* nminorkupper, int* idx_h2o,
        int* gpoint_flavor, int* band_lims_gpt,
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
    Stage S;
    const size_t n = sz(*ncol, *nlay), nf = n*size_t(*nflav);
    const int nl = *nminorlower, nu = *nminorupper;
    const int* d_gf = S.in(gpoint_flavor, sz(2, *ngpt)); const int* d_bl = S.in(band_lims_gpt, sz(2, *nband));
    const Float* d_kmajor = S.in(kmajor, sz(*ntemp, *neta, *npres + 1, *ngpt));
    const Float* d_kml = S.in(kminor_lower, sz(*ntemp, *neta, *nminorklower)); const Float* d_kmu = S.in(kminor_upper, sz(*ntemp, *neta, *nminorkupper));
    const int* d_mll = S.in(minor_limits_gpt_lower, sz(2, nl)); const int* d_mlu = S.in(minor_limits_gpt_upper, sz(2, nu));
    const Bool* d_sdl = S.in(minor_scales_with_density_lower, nl); const Bool* d_sdu = S.in(minor_scales_with_density_upper, nu);
    const Bool* d_scl = S.in(scale_by_complement_lower, nl); const Bool* d_scu = S.in(scale_by_complement_upper, nu);
    const int* d_iml = S.in(idx_minor_lower, nl); const int* d_imu = S.in(idx_minor_upper, nu);
    const int* d_isl = S.in(idx_minor_scaling_lower, nl); const int* d_isu = S.in(idx_minor_scaling_upper, nu);
    const int* d_ksl = S.in(kminor_start_lower, nl); const int* d_ksu = S.in(kminor_start_upper, nu);
    const Bool* d_tropo = S.in(tropo, n);
    const Float* d_colmix = S.in(col_mix, 2*nf); const Float* d_fmajor = S.in(fmajor, 8*nf); const Float* d_fminor = S.in(fminor, 4*nf);
    const Float* d_play = S.in(play, n); const Float* d_tlay = S.in(tlay, n); const Float* d_colgas = S.in(col_gas, n*size_t(*ngas + 1));
    const int* d_jeta = S.in(jeta, 2*nf); const int* d_jtemp = S.in(jtemp, n); const int* d_jpress = S.in(jpress, n);
    Float* d_tau = S.inout(tau, n*size_t(*ngpt));          // the kernels ADD onto tau (the caller zeroes it: Gas_optics_rrtmgp.cpp:1244)
    RRX_K(rrx_compute_tau_absorption, *ncol, *nlay, *nband, *ngpt, *ngas, *nflav, *neta, *npres, *ntemp,
          nl, *nminorklower, nu, *nminorkupper, *idx_h2o, d_gf, d_bl, d_kmajor, d_kml, d_kmu, d_mll, d_mlu, d_sdl, d_sdu, d_scl, d_scu,
          d_iml, d_imu, d_isl, d_isu, d_ksl, d_ksu, d_tropo, d_colmix, d_fmajor, d_fminor, d_play, d_tlay, d_colgas,
          d_jeta, d_jtemp, d_jpress, d_tau);
    S.finish();
}

void reorder_123x321_kernel(int* dim1, int* dim2, int* dim3, Float* array, Float* array_out)
{
    Stage S;
    const size_t n = sz(*dim1, *dim2, *dim3);
    const Float* a = S.in(array, n);
    Float* o = S.out(array_out, n);
    // the device launcher names the dimensions of its OUTPUT (ni fastest there; gas_optics_rrtmgp_kernels.cu:76-90), Fortran those of the input
    RRX_K(rrx_reorder123x321, *dim3, *dim2, *dim1, a, o);
    S.finish();
}

// (ngpt,nlay,ncol) absorption and Rayleigh optical depths -> tau, ssa, g as (ncol,nlay,ngpt): the pre-v1.5 Fortran kernel the
// reference header still declares (no caller in src/): transposes, then combine_abs_and_rayleigh
void combine_and_reorder_2str(int* ncol, int* nlay, int* ngpt, Float* tau_local, Float* tau_rayleigh, Float* tau, Float* ssa, Float* g)
{
    Stage S;
    const size_t n = sz(*ncol, *nlay, *ngpt);
    const Float* a = S.in(tau_local, n); const Float* r = S.in(tau_rayleigh, n);
    Float* at = S.alloc<Float>(n); Float* rt = S.alloc<Float>(n);
    RRX_K(rrx_reorder123x321, *ncol, *nlay, *ngpt, a, at);
    RRX_K(rrx_reorder123x321, *ncol, *nlay, *ngpt, r, rt);
    Float* d_tau = S.out(tau, n); Float* d_ssa = S.out(ssa, n); Float* d_g = S.out(g, n);
    RRX_K(rrx_combine_abs_and_rayleigh, *ncol, *nlay, *ngpt, at, rt, d_tau, d_ssa, d_g);
    S.finish();
}

void rrtmgp_compute_Planck_source(
        int* ncol, int* nlay, int* nbnd, int* ngpt, int* nflav, int* neta, int* npres, int* ntemp, int* nPlanckTemp,
        Float* tlay, Float* tlev, Float* tsfc, int* sfc_lay,
        Float* fmajor, int* jeta, Bool* tropo, int* jtemp, int* jpress,
        int* gpoint_bands, int* band_lims_gpt, Float* pfracin, Float* temp_ref_min,
        Float* totplnk_delta, Float* totplnk, int* gpoint_flavor,
        Float* sfc_src, Float* lay_src, Float* lev_src, Float* sfc_src_jac)
{
    Stage S;
    const size_t n = sz(*ncol, *nlay), nf = n*size_t(*nflav);
    const Float* d_tlay = S.in(tlay, n); const Float* d_tlev = S.in(tlev, sz(*ncol, *nlay + 1)); const Float* d_tsfc = S.in(tsfc, *ncol);
    const Float* d_fmajor = S.in(fmajor, 8*nf); const int* d_jeta = S.in(jeta, 2*nf);
    const Bool* d_tropo = S.in(tropo, n); const int* d_jtemp = S.in(jtemp, n); const int* d_jpress = S.in(jpress, n);
    const int* d_gb = S.in(gpoint_bands, *ngpt); const int* d_bl = S.in(band_lims_gpt, sz(2, *nbnd));
    const Float* d_pfrac = S.in(pfracin, sz(*ntemp, *neta, *npres + 1, *ngpt));
    const Float* d_totplnk = S.in(totplnk, sz(*nPlanckTemp, *nbnd)); const int* d_gf = S.in(gpoint_flavor, sz(2, *ngpt));
    Float* d_sfc = S.out(sfc_src, sz(*ncol, *ngpt)); Float* d_lay = S.out(lay_src, n*size_t(*ngpt));
    Float* d_lev = S.out(lev_src, sz(*ncol, *nlay + 1, *ngpt)); Float* d_jac = S.out(sfc_src_jac, sz(*ncol, *ngpt));
    RRX_K(rrx_compute_planck_source, *ncol, *nlay, *nbnd, *ngpt, *nflav, *neta, *npres, *ntemp, *nPlanckTemp,
          d_tlay, d_tlev, d_tsfc, *sfc_lay, d_fmajor, d_jeta, d_tropo, d_jtemp, d_jpress, d_gb, d_bl, d_pfrac,
          *temp_ref_min, *totplnk_delta, d_totplnk, d_gf, d_sfc, d_lay, d_lev, d_jac);
    S.finish();
}

void rrtmgp_compute_tau_rayleigh(
        int* ncol, int* nlay, int* nband, int* ngpt, int* ngas, int* nflav, int* neta, int* npres, int* ntemp,
        int* gpoint_flavor, int* band_lims_gpt, Float* krayl,
        int* idx_h2o, Float* col_dry, Float* col_gas, Float* fminor, int* eta, Bool* tropo, int* jtemp,
        Float* tau_rayleigh)
{
    Stage S;
    const size_t n = sz(*ncol, *nlay), nf = n*size_t(*nflav);
    const int* d_gf = S.in(gpoint_flavor, sz(2, *ngpt)); const int* d_bl = S.in(band_lims_gpt, sz(2, *nband));
    const Float* d_krayl = S.in(krayl, sz(*ntemp, *neta, *ngpt, 2));
    const Float* d_coldry = S.in(col_dry, n); const Float* d_colgas = S.in(col_gas, n*size_t(*ngas + 1));
    const Float* d_fminor = S.in(fminor, 4*nf); const int* d_jeta = S.in(eta, 2*nf);
    const Bool* d_tropo = S.in(tropo, n); const int* d_jtemp = S.in(jtemp, n);
    Float* d_tau = S.out(tau_rayleigh, n*size_t(*ngpt));
    RRX_K(rrx_compute_tau_rayleigh, *ncol, *nlay, *nband, *ngpt, *ngas, *nflav, *neta, *npres, *ntemp, d_gf, d_bl, d_krayl,
          *idx_h2o, d_coldry, d_colgas, d_fminor, d_jeta, d_tropo, d_jtemp, d_tau);
    S.finish();
}

// ---------------------------------------------------------------- solvers (src/Rte_lw.cpp:97, src/Rte_sw.cpp:111)
void rte_lw_solver_noscat(
        const int& ncol, const int& nlay, const int& ngpt, const Bool& top_at_1, const int& n_quad_angs,
        const Float* secants, const Float* gauss_wts_subset,
        const Float* tau, const Float* lay_source, const Float* lev_source,
        const Float* sfc_emis_gpt, const Float* sfc_source, const Float* inc_flux_diffuse,
        Float* gpt_flux_up, Float* gpt_flux_dn,
        const Bool& do_broadband, Float* flux_up_loc, Float* flux_dn_loc,
        const Bool& do_jacobians, const Float* sfc_source_jac, Float* gpt_flux_up_jac,
        const Bool& do_rescaling, const Float* /*ssa*/, const Float* /*g*/)
{
    if (do_rescaling) fail("rte_lw_solver_noscat: do_rescaling is not served (the reference's callers pass false, src/Rte_lw.cpp:186)");
    Stage S;
    const size_t n = sz(ncol, nlay), ng = sz(ncol, ngpt), nlev = sz(ncol, nlay + 1);
    const Float* d_sec = S.in(secants, ng*size_t(n_quad_angs)); const Float* d_wts = S.in(gauss_wts_subset, n_quad_angs);
    const Float* d_tau = S.in(tau, n*size_t(ngpt)); const Float* d_lay = S.in(lay_source, n*size_t(ngpt));
    const Float* d_lev = S.in(lev_source, nlev*size_t(ngpt));
    const Float* d_emis = S.in(sfc_emis_gpt, ng); const Float* d_src = S.in(sfc_source, ng); const Float* d_inc = S.in(inc_flux_diffuse, ng);
    // the reference hands the same arrays as per-g-point and broadband outputs (src/Rte_lw.cpp:199): touch only the live pair
    Float *d_up = nullptr, *d_dn = nullptr, *d_bup = nullptr, *d_bdn = nullptr;
    if (do_broadband) { d_bup = S.out(flux_up_loc, nlev); d_bdn = S.out(flux_dn_loc, nlev); }
    else              { d_up = S.out(gpt_flux_up, nlev*size_t(ngpt)); d_dn = S.out(gpt_flux_dn, nlev*size_t(ngpt)); }
    const Float* d_sjac = do_jacobians ? S.in(sfc_source_jac, ng) : nullptr;
    // the Jacobian output is sized like the live flux output (src/Rte_lw.cpp:181): (ncol, nlev, 1) in broadband mode -- the per-g-point
    // Jacobian then stays on the device and its g-point sum goes back (copying nlev*ngpt values overran the caller's array)
    Float *d_jac = nullptr, *d_jac_bb = nullptr;
    if (do_jacobians)
    {
        if (do_broadband) { d_jac = S.alloc<Float>(nlev*size_t(ngpt)); d_jac_bb = S.out(gpt_flux_up_jac, nlev); }
        else d_jac = S.out(gpt_flux_up_jac, nlev*size_t(ngpt));
    }
    RRX_K(rrx_lw_solver_noscat, ncol, nlay, ngpt, top_at_1, n_quad_angs, d_sec, d_wts, d_tau, d_lay, d_lev, d_emis, d_src, d_inc,
          d_up, d_dn, do_broadband, d_bup, d_bdn, do_jacobians, d_sjac, d_jac);
    if (d_jac_bb != nullptr) RRX_K(rrx_sum_broadband, ncol, nlay + 1, ngpt, d_jac, d_jac_bb);
    S.finish();
}

void rte_sw_solver_2stream(
        const int& ncol, const int& nlay, const int& ngpt, const Bool& top_at_1,
        const Float* tau, const Float* ssa, const Float* g, const Float* mu0,
        const Float* sfc_alb_dir_gpt, const Float* sfc_alb_dif_gpt, const Float* inc_flux_dir,
        Float* gpt_flux_up, Float* gpt_flux_dn, Float* gpt_flux_dir,
        const Bool& has_dif_bc, const Float* inc_flux_dif,
        const Bool& do_broadband, Float* flux_up_loc, Float* flux_dn_loc, Float* flux_dir_loc)
{
    // mu0 arrives as (ncol,nlay), a copy of mu0(ncol) per layer (src/Rte_sw.cpp:160-163); the device layer takes mu0(ncol)
    for (int ilay=1; ilay<nlay; ++ilay)
        if (std::memcmp(mu0, mu0 + size_t(ilay)*ncol, size_t(ncol)*sizeof(Float)) != 0)
            fail("rte_sw_solver_2stream: mu0 varying with height is not served (the device layer keeps mu0(ncol))");
    Stage S;
    const size_t n = sz(ncol, nlay, ngpt), ng = sz(ncol, ngpt), nlev = sz(ncol, nlay + 1);
    const Float* d_tau = S.in(tau, n); const Float* d_ssa = S.in(ssa, n); const Float* d_g = S.in(g, n);
    const Float* d_mu0 = S.in(mu0, ncol);
    const Float* d_adir = S.in(sfc_alb_dir_gpt, ng); const Float* d_adif = S.in(sfc_alb_dif_gpt, ng); const Float* d_inc = S.in(inc_flux_dir, ng);
    const Float* d_incdif = has_dif_bc ? S.in(inc_flux_dif, ng) : nullptr;
    Float *d_up = nullptr, *d_dn = nullptr, *d_dir = nullptr, *d_bup = nullptr, *d_bdn = nullptr, *d_bdir = nullptr;
    if (do_broadband) { d_bup = S.out(flux_up_loc, nlev); d_bdn = S.out(flux_dn_loc, nlev); d_bdir = S.out(flux_dir_loc, nlev); }
    else { d_up = S.out(gpt_flux_up, nlev*size_t(ngpt)); d_dn = S.out(gpt_flux_dn, nlev*size_t(ngpt)); d_dir = S.out(gpt_flux_dir, nlev*size_t(ngpt)); }
    RRX_K(rrx_sw_solver_2stream, ncol, nlay, ngpt, top_at_1, d_tau, d_ssa, d_g, d_mu0, d_adir, d_adif, d_inc, d_up, d_dn, d_dir,
          has_dif_bc, d_incdif, do_broadband, d_bup, d_bdn, d_bdir);
    S.finish();
}

// ---------------------------------------------------------------- optical properties (src/Optical_props.cpp:154-200)
void rte_increment_2stream_by_2stream(
        int* ncol, int* nlev, int* ngpt, Float* tau_inout, Float* ssa_inout, Float* g_inout, Float* tau_in, Float* ssa_in, Float* g_in)
{
    Stage S;
    const size_t n = sz(*ncol, *nlev, *ngpt);
    Float* t = S.inout(tau_inout, n); Float* w = S.inout(ssa_inout, n); Float* gg = S.inout(g_inout, n);
    const Float* t2 = S.in(tau_in, n); const Float* w2 = S.in(ssa_in, n); const Float* g2 = S.in(g_in, n);
    RRX_K(rrx_increment_2stream_by_2stream, *ncol, *nlev, *ngpt, t, w, gg, t2, w2, g2);
    S.finish();
}

void rte_increment_1scalar_by_1scalar(int* ncol, int* nlev, int* ngpt, Float* tau_inout, Float* tau_in)
{
    Stage S;
    const size_t n = sz(*ncol, *nlev, *ngpt);
    Float* t = S.inout(tau_inout, n); const Float* t2 = S.in(tau_in, n);
    RRX_K(rrx_increment_1scalar_by_1scalar, *ncol, *nlev, *ngpt, t, t2);
    S.finish();
}

void rte_inc_2stream_by_2stream_bybnd(
        int* ncol, int* nlev, int* ngpt, Float* tau_inout, Float* ssa_inout, Float* g_inout,
        Float* tau_in, Float* ssa_in, Float* g_in, int* nbnd, int* band_lims_gpoint)
{
    Stage S;
    const size_t n = sz(*ncol, *nlev, *ngpt), nb = sz(*ncol, *nlev, *nbnd);
    Float* t = S.inout(tau_inout, n); Float* w = S.inout(ssa_inout, n); Float* gg = S.inout(g_inout, n);
    const Float* t2 = S.in(tau_in, nb); const Float* w2 = S.in(ssa_in, nb); const Float* g2 = S.in(g_in, nb);
    const int* lims = S.in(band_lims_gpoint, sz(2, *nbnd));
    RRX_K(rrx_inc_2stream_by_2stream_bybnd, *ncol, *nlev, *ngpt, t, w, gg, t2, w2, g2, *nbnd, lims);
    S.finish();
}

void rte_inc_1scalar_by_1scalar_bybnd(int* ncol, int* nlev, int* ngpt, Float* tau_inout, Float* tau_in, int* nbnd, int* band_lims_gpoint)
{
    Stage S;
    Float* t = S.inout(tau_inout, sz(*ncol, *nlev, *ngpt)); const Float* t2 = S.in(tau_in, sz(*ncol, *nlev, *nbnd));
    const int* lims = S.in(band_lims_gpoint, sz(2, *nbnd));
    RRX_K(rrx_inc_1scalar_by_1scalar_bybnd, *ncol, *nlev, *ngpt, t, t2, *nbnd, lims);
    S.finish();
}

void rte_delta_scale_2str_k(int* ncol, int* nlev, int* ngpt, Float* tau_inout, Float* ssa_inout, Float* g_inout)
{
    Stage S;
    const size_t n = sz(*ncol, *nlev, *ngpt);
    Float* t = S.inout(tau_inout, n); Float* w = S.inout(ssa_inout, n); Float* gg = S.inout(g_inout, n);
    RRX_K(rrx_delta_scale_2str_k, *ncol, *nlev, *ngpt, t, w, gg);
    S.finish();
}
}
}
