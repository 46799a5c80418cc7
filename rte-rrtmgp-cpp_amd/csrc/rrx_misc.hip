// Element-wise / reduction / gather kernels of the hot path and the small runtime of librrx_hip.so.
// Replaces Optical_props_kernels_cuda::*, Fluxes_kernels_cuda::*, Subset_kernels_cuda::* and the inline kernels of
// src_cuda/{Gas_optics_rrtmgp,Rte_lw,Rte_sw,Cloud_optics}.cu and include/Array.h (file:line per function in
// include/rrx_hip.h). All of these are pure HBM-bandwidth work: every kernel keeps the column index on the lanes
// (the reference's optical-props launchers put the g-point on threadIdx.x, i.e. strided by ncol*nlay) and moves
// 16 B per lane where the alignment allows.
#include <mutex>
#include <vector>
#include <cstdlib>
#include "rrx_common.h"
#include "rrx_hip.h"

namespace rrx
{
    static thread_local std::string g_last_error;
    void set_error(const std::string& msg) { g_last_error = msg; }
    Tuning& tuning()
    {
        static thread_local Tuning t = []
        {
            Tuning d;
            if (const char* e = getenv("RRX_SYNC")) d.sync_waves = atoi(e);
            if (const char* e = getenv("RRX_GO_SHARE")) d.go_share = atoi(e);
            if (const char* e = getenv("RRX_GO_WINDOW")) d.go_window = atoi(e);
            return d;
        }();
        return t;
    }
    namespace
    {
        struct WsSlot { int dev; hipStream_t st; void* p; size_t cap; };
        std::vector<WsSlot>& ws_slots() { static thread_local std::vector<WsSlot> slots; return slots; }
        size_t ws_keep_limit()
        {
            static const size_t limit = []
            {
                if (const char* e = std::getenv("RRX_WORKSPACE_KEEP")) return size_t(std::strtoull(e, nullptr, 10));
                return size_t(32) << 30;
            }();
            return limit;
        }
        WsSlot* ws_find(hipStream_t st)
        {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess) return nullptr;
            for (WsSlot& s : ws_slots()) if (s.dev == dev && s.st == st) return &s;
            return nullptr;
        }
    }
    void* cached_workspace(hipStream_t st, const size_t bytes)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) throw std::runtime_error("workspace allocation failed");
        WsSlot* slot = ws_find(st);
        if (slot == nullptr) { ws_slots().push_back(WsSlot{dev, st, nullptr, 0}); slot = &ws_slots().back(); }
        if (slot->cap < bytes)
        {
            if (slot->p != nullptr) (void)hipFreeAsync(slot->p, st);
            slot->p = nullptr; slot->cap = 0;
            keep_pool_memory();
            if (hipMallocAsync(&slot->p, bytes, st) != hipSuccess) throw std::runtime_error("workspace allocation failed");
            slot->cap = bytes;
        }
        return slot->p;
    }
    void release_workspace(hipStream_t st)
    {
        WsSlot* slot = ws_find(st);
        if (slot == nullptr) return;
        if (slot->p != nullptr) (void)hipFreeAsync(slot->p, st);
        ws_slots().erase(ws_slots().begin() + (slot - ws_slots().data()));
    }
    void trim_workspace(hipStream_t st)
    {
        const WsSlot* slot = ws_find(st);
        if (slot != nullptr && slot->cap > ws_keep_limit()) release_workspace(st);
    }
    size_t workspace_bytes(hipStream_t st)
    {
        const WsSlot* slot = ws_find(st);
        return slot ? slot->cap : 0;
    }
    int check_launch(const char* what)
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess)
        {
            set_error(std::string(what) + ": " + hipGetErrorString(e));
            return 2;
        }
        return 0;
    }
}

namespace
{
using namespace rrx;

inline int grid1d(const size_t n) { return int(std::min<size_t>((n + 255)/256, 256*16)); }
#define RRX_GRID_STRIDE(i, n) for (size_t i = size_t(blockIdx.x)*blockDim.x + threadIdx.x; i < (n); i += size_t(gridDim.x)*blockDim.x)

// heating rate of a layer from the net (down - up) broadband flux at its two levels: dT/dt = -(g/cp) d(F_dn - F_up)/dp [K/s]
// (no counterpart in the reference library: its host model MicroHH differentiates the fluxes itself)
template<typename F>
__global__ void heating_rate_kernel(const size_t n, const int ncol, const F g_over_cp, const F* __restrict__ flux_net,
                                    const F* __restrict__ plev, F* __restrict__ hr)
{
    RRX_GRID_STRIDE(i, n)
    {
        const F dp = plev[i + ncol] - plev[i];
        hr[i] = -g_over_cp * (flux_net[i + ncol] - flux_net[i]) / dp;
    }
}

// ---- optical props: /root/reference/src_kernels_cuda/optical_props_kernels.cu:31-161 ----
template<typename F>
__global__ void inc_1scl_kernel(const size_t n, F* __restrict__ tau1, const F* __restrict__ tau2)
{
    RRX_GRID_STRIDE(i, n) tau1[i] = tau1[i] + tau2[i];
}

template<typename F>
__device__ __forceinline__ void inc_2str(F& tau1, F& ssa1, F& g1, const F tau2, const F ssa2, const F g2, const F eps)
{
    const F tau12 = tau1 + tau2;
    const F tauscat12 = (tau1 * ssa1) + (tau2 * ssa2);
    g1 = ((tau1 * ssa1 * g1) + (tau2 * ssa2 * g2)) / max(tauscat12, eps);
    ssa1 = tauscat12 / max(eps, tau12);
    tau1 = tau12;
}

template<typename F>
__global__ void inc_2str_kernel(const size_t n, const F eps, F* __restrict__ tau1, F* __restrict__ ssa1, F* __restrict__ g1,
        const F* __restrict__ tau2, const F* __restrict__ ssa2, const F* __restrict__ g2)
{
    RRX_GRID_STRIDE(i, n)
    {
        F t = tau1[i], w = ssa1[i], gg = g1[i];
        inc_2str(t, w, gg, tau2[i], ssa2[i], g2[i], eps);
        tau1[i] = t; ssa1[i] = w; g1[i] = gg;
    }
}

// g-point -> band map resolved once per block row (blockIdx.y = g-point), cells on the lanes
template<typename F>
__global__ void inc_1scl_bybnd_kernel(const size_t ncl, F* __restrict__ tau1, const F* __restrict__ tau2,
        const int nbnd, const int* __restrict__ band_lims_gpt)
{
    const int igpt = blockIdx.y;
    for (int ibnd=0; ibnd<nbnd; ++ibnd)
        if (igpt+1 >= band_lims_gpt[2*ibnd] && igpt+1 <= band_lims_gpt[2*ibnd+1])
        {
            RRX_GRID_STRIDE(i, ncl) tau1[i + igpt*ncl] = tau1[i + igpt*ncl] + tau2[i + ibnd*ncl];
        }
}

template<typename F>
__global__ void inc_2str_bybnd_kernel(const size_t ncl, const F eps, F* __restrict__ tau1, F* __restrict__ ssa1, F* __restrict__ g1,
        const F* __restrict__ tau2, const F* __restrict__ ssa2, const F* __restrict__ g2,
        const int nbnd, const int* __restrict__ band_lims_gpt)
{
    const int igpt = blockIdx.y;
    for (int ibnd=0; ibnd<nbnd; ++ibnd)
        if (igpt+1 >= band_lims_gpt[2*ibnd] && igpt+1 <= band_lims_gpt[2*ibnd+1])
        {
            RRX_GRID_STRIDE(i, ncl)
            {
                const size_t o = i + igpt*ncl, b = i + ibnd*ncl;
                F t = tau1[o], w = ssa1[o], gg = g1[o];
                inc_2str(t, w, gg, tau2[b], ssa2[b], g2[b], eps);
                tau1[o] = t; ssa1[o] = w; g1[o] = gg;
            }
        }
}

template<typename F>
__global__ void delta_scale_kernel(const size_t n, const F eps, F* __restrict__ tau, F* __restrict__ ssa, F* __restrict__ g)
{
    RRX_GRID_STRIDE(i, n)
    {
        const F gv = g[i], wv = ssa[i];
        const F f = gv * gv;
        const F wf = wv * f;
        tau[i] *= (F(1.) - wf);
        ssa[i] = (wv - wf) / max(eps, F(1.) - wf);
        g[i] = (gv - f) / max(eps, F(1.) - f);
    }
}

// ---- fluxes: /root/reference/src_kernels_cuda/fluxes_kernels.cu:27-62; by-band: mo_fluxes_byband_kernels.F90 ----
template<typename F>
__global__ void sum_broadband_kernel(const size_t nlc, const int ngpt, const F* __restrict__ in, F* __restrict__ out)
{
    RRX_GRID_STRIDE(i, nlc)
    {
        F s = F(0.);
        for (int ig=0; ig<ngpt; ++ig) s += in[i + size_t(ig)*nlc];
        out[i] = s;
    }
}

template<typename F>
__global__ void net_kernel(const size_t n, const F* __restrict__ dn, const F* __restrict__ up, F* __restrict__ net)
{
    RRX_GRID_STRIDE(i, n) net[i] = dn[i] - up[i];
}

template<typename F, bool NET>
__global__ void byband_kernel(const size_t nlc, const int* __restrict__ band_lims, const F* __restrict__ a, const F* __restrict__ b, F* __restrict__ out)
{
    const int ibnd = blockIdx.y;
    const int g0 = band_lims[2*ibnd]-1, g1 = band_lims[2*ibnd+1]-1;
    RRX_GRID_STRIDE(i, nlc)
    {
        F s = NET ? a[i + size_t(g0)*nlc] - b[i + size_t(g0)*nlc] : a[i + size_t(g0)*nlc];
        for (int ig=g0+1; ig<=g1; ++ig)
            s += NET ? a[i + size_t(ig)*nlc] - b[i + size_t(ig)*nlc] : a[i + size_t(ig)*nlc];
        out[i + size_t(ibnd)*nlc] = s;
    }
}

// ---- subset scatter: /root/reference/src_kernels_cuda/subset_kernels.cu:2-98 ----
template<typename F> struct PtrPack { F* full[4]; const F* sub[4]; };

template<typename F>
__global__ void get_from_subset_kernel(const int ncol, const size_t nrest, const int ncol_in, const int col_s_in, const int narr, const PtrPack<F> p)
{
    const size_t n = size_t(ncol_in)*nrest;
    RRX_GRID_STRIDE(i, n)
    {
        const size_t ic = i % ncol_in, r = i / ncol_in;
        const size_t o = ic + col_s_in - 1 + r*ncol;
        for (int a=0; a<narr; ++a) p.full[a][o] = p.sub[a][i];
    }
}

// ---- host-class helper kernels ----
template<typename F>
__global__ void fill_gases_kernel(const int ncol, const int nlay, const int dim1, const int dim2, const int igas,
        F* __restrict__ vmr_out, const F* __restrict__ vmr_in, F* __restrict__ col_gas, const F* __restrict__ col_dry)
{
    const size_t ncl = size_t(ncol)*nlay;
    RRX_GRID_STRIDE(i, ncl)
    {
        if (igas > 0)
        {
            F v;
            if (dim1 == 1 && dim2 == 1) v = vmr_in[0];
            else if (dim1 == 1)         v = vmr_in[i / ncol];
            else                        v = vmr_in[i];
            vmr_out[i + size_t(igas-1)*ncl] = v;
            col_gas[i + size_t(igas)*ncl] = v * col_dry[i];
        }
        else
            col_gas[i] = col_dry[i];
    }
}

// all gases in one launch (the product chain: the per-gas vmr copy of fill_gases_kernel is not kept)
constexpr int RRX_MAX_GASES = 32;
template<typename F> struct GasTable { const F* vmr[RRX_MAX_GASES]; int dim1[RRX_MAX_GASES], dim2[RRX_MAX_GASES]; };

template<typename F>
__global__ void fill_gases_all_kernel(const int ncol, const int nlay, const int ngas, const GasTable<F> gt,
                                      F* __restrict__ col_gas, const F* __restrict__ col_dry)
{
    const size_t ncl = size_t(ncol)*nlay;
    RRX_GRID_STRIDE(i, ncl)
    {
        const F cd = col_dry[i];
        col_gas[i] = cd;
        for (int igas=0; igas<ngas; ++igas)
        {
            const F* __restrict__ src = gt.vmr[igas];
            const F v = (gt.dim1[igas] == 1 && gt.dim2[igas] == 1) ? src[0] : (gt.dim1[igas] == 1 ? src[i / ncol] : src[i]);
            col_gas[i + size_t(igas+1)*ncl] = v * cd;
        }
    }
}

template<typename F>
__global__ void col_dry_kernel(const int ncol, const int nlay, const F* __restrict__ vmr_h2o, const F* __restrict__ plev, F* __restrict__ col_dry)
{
    constexpr F g0 = 9.80665;
    constexpr F avogad = 6.02214076e23;
    constexpr F m_dry = 0.028964;
    constexpr F m_h2o = 0.018016;
    const size_t ncl = size_t(ncol)*nlay;
    RRX_GRID_STRIDE(i, ncl)
    {
        const F delta_plev = abs(plev[i] - plev[i + ncol]);
        const F h = vmr_h2o[i];
        const F m_air = (m_dry + m_h2o * h) / (F(1.) + h);
        F cd = F(10.) * delta_plev * avogad / (F(1000.)*m_air*F(100.)*g0);
        cd /= (F(1.) + h);
        col_dry[i] = cd;
    }
}

template<typename F>
__global__ void expand_and_transpose_kernel(const int ncol, const int nbnd, const int* __restrict__ limits, const F* __restrict__ in, F* __restrict__ out)
{
    const int ibnd = blockIdx.y;
    const int g0 = limits[2*ibnd]-1, g1 = limits[2*ibnd+1];
    RRX_GRID_STRIDE(icol, size_t(ncol))
    {
        const F v = in[ibnd + icol*nbnd];
        for (int ig=g0; ig<g1; ++ig) out[icol + size_t(ig)*ncol] = v;
    }
}

// (grid.y = g-point: no 64-bit division per element -- the flat-index forms of rounds 1-3 took 8 and 15 us for 2 048 x 256 values)
template<typename F>
__global__ void spread_col_kernel(const int ncol, const int ngpt, F* __restrict__ out, const F* __restrict__ src)
{
    const F v = src[blockIdx.y];
    for (int icol = blockIdx.x*blockDim.x + threadIdx.x; icol < ncol; icol += gridDim.x*blockDim.x) out[icol + size_t(blockIdx.y)*ncol] = v;
}

template<typename F>
__global__ void scale_cols_kernel(const int ncol, const int ngpt, F* __restrict__ out, const F* __restrict__ fac)
{
    for (int icol = blockIdx.x*blockDim.x + threadIdx.x; icol < ncol; icol += gridDim.x*blockDim.x) out[icol + size_t(blockIdx.y)*ncol] *= fac[icol];
}

// spread_col followed by scaling_to_subset in one pass: toa_src(icol, igpt) = solar_source(igpt) * tsi_scaling(icol), rounded like
// the two kernels (one multiplication)
template<typename F>
__global__ void toa_source_kernel(const int ncol, const int ngpt, F* __restrict__ out, const F* __restrict__ src, const F* __restrict__ fac)
{
    const F v = src[blockIdx.y];
    for (int icol = blockIdx.x*blockDim.x + threadIdx.x; icol < ncol; icol += gridDim.x*blockDim.x) out[icol + size_t(blockIdx.y)*ncol] = v * fac[icol];
}

// /root/reference/src/Cloud_optics.cpp:72-107 (and src_cuda/Cloud_optics.cu:31-70)
template<typename F>
__device__ __forceinline__ void cloud_from_table(const F cwp, const F re, const int nsteps, const F step_size, const F offset,
        const F* __restrict__ tau_table, const F* __restrict__ ssa_table, const F* __restrict__ asy_table,
        F& tau, F& taussa, F& taussag)
{
    if (cwp > F(0.))
    {
        const int index = min(int((re - offset) / step_size)+1, nsteps-1);
        const F fint = (re - offset) / step_size - (index-1);
        tau = cwp * (tau_table[index-1] + fint * (tau_table[index] - tau_table[index-1]));
        taussa = tau * (ssa_table[index-1] + fint * (ssa_table[index] - ssa_table[index-1]));
        taussag = taussa * (asy_table[index-1] + fint * (asy_table[index] - asy_table[index-1]));
    }
    else { tau = F(0.); taussa = F(0.); taussag = F(0.); }
}

// One thread per (column, layer): water paths and particle sizes are read once, the table position (it depends on the size only) is
// found once, and the thread walks over the bands (round 4; a thread per (cell, band) re-read the four inputs and redid the two
// divisions nbnd times). DELTA: delta_scale_2str_k (optical_props_kernels.cu:103-136) applied to the values before they are stored --
// the reference's cloud_optics() + delta_scale() pair (Radiation_solver.cu:773-792) in one pass, same expressions, same bits.
template<typename F, bool TWOSTR, bool DELTA>
__global__ void cloud_optics_kernel(const size_t ncl, const int nbnd, const int nsize_liq, const int nsize_ice,
        const F radliq_lwr, const F liq_step, const F diamice_lwr, const F ice_step,
        const F* __restrict__ lut_extliq, const F* __restrict__ lut_ssaliq, const F* __restrict__ lut_asyliq,
        const F* __restrict__ lut_extice, const F* __restrict__ lut_ssaice, const F* __restrict__ lut_asyice,
        const F* __restrict__ clwp, const F* __restrict__ ciwp, const F* __restrict__ reliq, const F* __restrict__ deice,
        F* __restrict__ tau, F* __restrict__ ssa, F* __restrict__ g)
{
    const F eps = Lim<F>::eps();
    const F eps_delta = Lim<F>::tiny()*F(3.);
    RRX_GRID_STRIDE(i, ncl)
    {
        const F lw = clwp[i], iw = ciwp[i], rl = reliq[i], di = deice[i];
        // /root/reference/src/Cloud_optics.cpp:72-107: position in the size tables
        int il = 1, ii = 1; F fl = F(0.), fi = F(0.);
        if (lw > F(0.)) { il = min(int((rl - radliq_lwr) / liq_step)+1, nsize_liq-1); fl = (rl - radliq_lwr) / liq_step - (il-1); }
        if (iw > F(0.)) { ii = min(int((di - diamice_lwr) / ice_step)+1, nsize_ice-1); fi = (di - diamice_lwr) / ice_step - (ii-1); }
        for (int ibnd=0; ibnd<nbnd; ++ibnd)
        {
            F lt = F(0.), lts = F(0.), ltsg = F(0.), it = F(0.), its = F(0.), itsg = F(0.);
            if (lw > F(0.))
            {
                const F* te = lut_extliq + size_t(ibnd)*nsize_liq; const F* ts = lut_ssaliq + size_t(ibnd)*nsize_liq; const F* ta = lut_asyliq + size_t(ibnd)*nsize_liq;
                lt = lw * (te[il-1] + fl * (te[il] - te[il-1]));
                lts = lt * (ts[il-1] + fl * (ts[il] - ts[il-1]));
                ltsg = lts * (ta[il-1] + fl * (ta[il] - ta[il-1]));
            }
            if (iw > F(0.))
            {
                const F* te = lut_extice + size_t(ibnd)*nsize_ice; const F* ts = lut_ssaice + size_t(ibnd)*nsize_ice; const F* ta = lut_asyice + size_t(ibnd)*nsize_ice;
                it = iw * (te[ii-1] + fi * (te[ii] - te[ii-1]));
                its = it * (ts[ii-1] + fi * (ts[ii] - ts[ii-1]));
                itsg = its * (ta[ii-1] + fi * (ta[ii] - ta[ii-1]));
            }
            const size_t o = i + size_t(ibnd)*ncl;
            if constexpr (TWOSTR)
            {
                const F t = lt + it, tsc = lts + its, tsg = ltsg + itsg;
                F tv = t, wv = tsc / max(t, eps), gv = tsg / max(tsc, eps);
                if constexpr (DELTA)
                {
                    const F f = gv * gv;
                    const F wf = wv * f;
                    tv *= (F(1.) - wf);
                    const F w2 = (wv - wf) / max(eps_delta, F(1.) - wf);
                    gv = (gv - f) / max(eps_delta, F(1.) - f);
                    wv = w2;
                }
                tau[o] = tv; ssa[o] = wv; g[o] = gv;
            }
            else
                tau[o] = (lt - lts) + (it - its);
        }
    }
}

// ---- aerosol optics: /root/reference/src/Aerosol_optics.cpp:24-224, src_cuda/Aerosol_optics.cu:9-263 ----
// 11 CAMS species, each a (mixing ratio, table column) pair; hydrophilic species are looked up in the humidity class of the cell.
// One thread per (column, layer): humidity class, dp/g and the 11 mixing ratios are read once and reused for every band
// (the reference launches a thread per (column, layer, band) and re-reads them nbnd times; its three temporaries
// tau / tau*ssa / tau*ssa*g and the second, finalising kernel are folded into this one). Species are accumulated in the
// order of the CPU text (SS1 SS2 SS3 DU1 DU2 DU3 OM1 OM2 BC1 BC2 SU), which the oracle restates.
template<typename F> struct AerosolSpecies { const F* mmr[11]; int per_column[11]; };

// {index into aermr01..11, hydrophilic?, 0-based table column}: Aerosol_optics.cpp:58-157
__constant__ const signed char aerosol_species_table[11][3] = {
    {0, 1, 0}, {1, 1, 1}, {2, 1, 2},          // SS1 SS2 SS3 : aermr01..03, hydrophilic 1..3
    {3, 0, 0}, {4, 0, 7}, {5, 0, 5},          // DU1 DU2 DU3 : aermr04..06, hydrophobic 1, 8, 6
    {7, 0, 9}, {6, 1, 3},                     // OM1 (aermr08, hydrophobic 10), OM2 (aermr07, hydrophilic 4)
    {8, 0, 10}, {9, 0, 10},                   // BC1 BC2 : aermr09, aermr10, hydrophobic 11
    {10, 1, 4}};                              // SU : aermr11, hydrophilic 5

template<typename F>
__global__ void aerosol_optics_kernel(const int ncol, const int nlay, const int nbnd, const int nhum,
        const AerosolSpecies<F> sp, const F* __restrict__ rh, const F* __restrict__ plev, const F* __restrict__ rh_upper,
        const F* __restrict__ mext_phobic, const F* __restrict__ ssa_phobic, const F* __restrict__ g_phobic,
        const F* __restrict__ mext_philic, const F* __restrict__ ssa_philic, const F* __restrict__ g_philic,
        F* __restrict__ tau, F* __restrict__ ssa, F* __restrict__ g)
{
    const size_t ncl = size_t(ncol)*nlay;
    const F eps = Lim<F>::eps();
    RRX_GRID_STRIDE(i, ncl)
    {
        const int ilay = int(i / ncol), icol = int(i - size_t(ilay)*ncol);
        // first class whose upper bound reaches the cell's humidity (Aerosol_optics.cpp:24-36); the reference reads past the
        // table for rh above the last bound, here the last class is used
        const F rh_c = rh[i];
        int ihum = 0;
        while (ihum < nhum-1 && rh_upper[ihum] < rh_c) ++ihum;
        const F dpg = abs(plev[i] - plev[i + ncol]) / F(9.81);
        F od[11];
        int lut[11];
        #pragma unroll
        for (int s=0; s<11; ++s)
        {
            const int im = aerosol_species_table[s][0];
            const bool philic = aerosol_species_table[s][1];
            od[s] = sp.mmr[im][sp.per_column[im] ? i : size_t(ilay)] * dpg;
            lut[s] = philic ? (ihum + aerosol_species_table[s][2]*nhum)*nbnd : aerosol_species_table[s][2]*nbnd;
        }
        (void)icol;
        for (int ibnd=0; ibnd<nbnd; ++ibnd)
        {
            F t = F(0.), ts = F(0.), tsg = F(0.);
            #pragma unroll
            for (int s=0; s<11; ++s)
            {
                const bool philic = aerosol_species_table[s][1];
                const int k = lut[s] + ibnd;
                const F local_od = od[s] * (philic ? mext_philic[k] : mext_phobic[k]);
                const F w = philic ? ssa_philic[k] : ssa_phobic[k];
                const F a = philic ? g_philic[k] : g_phobic[k];
                t += local_od;
                ts += local_od * w;
                tsg += local_od * w * a;
            }
            const size_t o = i + size_t(ibnd)*ncl;
            tau[o] = t;
            ssa[o] = ts / max(t, eps);
            g[o] = tsg / max(ts, eps);
        }
    }
}

template<typename F>
__global__ void subset_cols_kernel(const int ncol_full, const size_t nrest, const int col_s, const int ncol_sub, const F* __restrict__ in, F* __restrict__ out)
{
    const size_t n = size_t(ncol_sub)*nrest;
    RRX_GRID_STRIDE(i, n)
    {
        const size_t ic = i % ncol_sub, r = i / ncol_sub;
        out[i] = in[ic + col_s - 1 + r*ncol_full];
    }
}

// Generic N-D block gather with broadcast of singleton dimensions: include/Array.h:311-350,579-622 of the reference.
struct SubsetND { int ndim; int sub_dims[7]; long long strides[7]; int starts[7]; int spread[7]; };

template<typename T>
__global__ void subset_nd_kernel(const SubsetND sd, const size_t n, const T* __restrict__ in, T* __restrict__ out)
{
    RRX_GRID_STRIDE(i, n)
    {
        size_t rem = i; long long src = 0;
        for (int d=0; d<sd.ndim; ++d)
        {
            const int id = int(rem % sd.sub_dims[d]); rem /= sd.sub_dims[d];
            src += (sd.spread[d] ? 0 : (long long)(id + sd.starts[d])) * sd.strides[d];
        }
        out[i] = in[src];
    }
}

template<typename F>
__global__ void fill_kernel(const size_t n, const F v, F* __restrict__ a)
{
    RRX_GRID_STRIDE(i, n) a[i] = v;
}
}  // namespace


extern "C"
{
const char* rrx_last_error(void) { return rrx::g_last_error.c_str(); }
int rrx_set_gas_window(int on) { rrx::tuning().go_window = on; return 0; }

#define RRX_HIP_OK(call, name) do { const hipError_t e_ = (call); if (e_ != hipSuccess) { \
    rrx::set_error(std::string(name) + ": " + hipGetErrorString(e_)); return 2; } } while (0)

int rrx_device_count(int* n) { RRX_HIP_OK(hipGetDeviceCount(n), "rrx_device_count"); return 0; }
int rrx_set_device(int dev) { RRX_HIP_OK(hipSetDevice(dev), "rrx_set_device"); return 0; }
int rrx_malloc(void** ptr, unsigned long long bytes) { RRX_HIP_OK(hipMalloc(ptr, bytes ? bytes : 1), "rrx_malloc"); return 0; }
int rrx_free(void* ptr) { RRX_HIP_OK(hipFree(ptr), "rrx_free"); return 0; }
// Stream-ordered allocation (the reference's memory-pool strategies, src_cuda/mem_pool_gpu.cu, in one line of HIP): memory comes
// from the device's default pool, whose release threshold is lifted once, so blocks freed by one solve are reused by the next
// without going back to the driver -- and without hipFree's device-wide synchronisation.
int rrx_malloc_async(void** ptr, unsigned long long bytes, void* stream)
{
    rrx::keep_pool_memory();      // freed blocks stay in the pool for the next solve (rrx_common.h)
    RRX_HIP_OK(hipMallocAsync(ptr, bytes ? bytes : 1, static_cast<hipStream_t>(stream)), "rrx_malloc_async");
    return 0;
}
int rrx_free_async(void* ptr, void* stream) { RRX_HIP_OK(hipFreeAsync(ptr, static_cast<hipStream_t>(stream)), "rrx_free_async"); return 0; }
// Release of a block that was allocated under one stream and may have been used under another (rrx_host::set_stream between the
// two): the release stream first waits for the work enqueued on the allocation stream, then the block returns to the pool in the
// release stream's order -- after every use on either stream. A failing stream-ordered free (e.g. a destroyed stream) falls back
// to hipFree, so the block is never leaked.
int rrx_free_async_ordered(void* ptr, void* alloc_stream, void* release_stream)
{
    if (ptr == nullptr) return 0;
    hipStream_t a = static_cast<hipStream_t>(alloc_stream), r = static_cast<hipStream_t>(release_stream);
    bool ok = true;
    if (a != r)
    {
        hipEvent_t ev;
        ok = hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
        if (ok)
        {
            ok = hipEventRecord(ev, a) == hipSuccess && hipStreamWaitEvent(r, ev, 0) == hipSuccess;
            (void)hipEventDestroy(ev);
        }
    }
    if (ok && hipFreeAsync(ptr, r) == hipSuccess) return 0;
    (void)hipGetLastError();
    RRX_HIP_OK(hipFree(ptr), "rrx_free_async_ordered");         // synchronises the device: correct, only slower
    return 0;
}
// host <-> device copies enqueued on `stream` and awaited (the host buffer is consumed / filled when the call returns)
int rrx_memcpy_h2d_stream(void* dst, const void* src, unsigned long long bytes, void* stream)
{
    RRX_HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)), "rrx_memcpy_h2d_stream");
    RRX_HIP_OK(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "rrx_memcpy_h2d_stream");
    return 0;
}
int rrx_memcpy_d2h_stream(void* dst, const void* src, unsigned long long bytes, void* stream)
{
    RRX_HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)), "rrx_memcpy_d2h_stream");
    RRX_HIP_OK(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "rrx_memcpy_d2h_stream");
    return 0;
}
int rrx_memcpy_h2d(void* dst, const void* src, unsigned long long bytes) { RRX_HIP_OK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), "rrx_memcpy_h2d"); return 0; }
int rrx_memcpy_d2h(void* dst, const void* src, unsigned long long bytes) { RRX_HIP_OK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), "rrx_memcpy_d2h"); return 0; }
int rrx_memcpy_d2d(void* dst, const void* src, unsigned long long bytes, void* stream) { RRX_HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)), "rrx_memcpy_d2d"); return 0; }
int rrx_memset(void* dst, int value, unsigned long long bytes, void* stream) { RRX_HIP_OK(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)), "rrx_memset"); return 0; }
int rrx_synchronize(void* stream) { RRX_HIP_OK(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "rrx_synchronize"); return 0; }
int rrx_stream_create(void** stream) { hipStream_t s; RRX_HIP_OK(hipStreamCreate(&s), "rrx_stream_create"); *stream = s; return 0; }
int rrx_stream_destroy(void* stream)
{
    rrx::release_workspace(static_cast<hipStream_t>(stream));       // the stream owns its workspace block (rrx_common.h)
    RRX_HIP_OK(hipStreamDestroy(static_cast<hipStream_t>(stream)), "rrx_stream_destroy");
    return 0;
}
int rrx_release_workspace(void* stream) { rrx::release_workspace(static_cast<hipStream_t>(stream)); return 0; }
unsigned long long rrx_workspace_bytes(void* stream) { return rrx::workspace_bytes(static_cast<hipStream_t>(stream)); }

#define ST static_cast<hipStream_t>(stream)

int rrx_subset_nd(void* out, const void* in, int elem_bytes, int ndim, const int* sub_dims, const long long* strides,
                  const int* starts, const int* spread, void* stream)
{
    RRX_TRY
    if (ndim < 1 || ndim > 7) throw std::runtime_error("ndim must be 1..7");
    SubsetND sd; sd.ndim = ndim; size_t n = 1;
    for (int d=0; d<ndim; ++d) { sd.sub_dims[d] = sub_dims[d]; sd.strides[d] = strides[d]; sd.starts[d] = starts[d]; sd.spread[d] = spread[d]; n *= sub_dims[d]; }
    if (n == 0) return 0;
    if (elem_bytes == 8) subset_nd_kernel<double><<<grid1d(n), 256, 0, ST>>>(sd, n, static_cast<const double*>(in), static_cast<double*>(out));
    else if (elem_bytes == 4) subset_nd_kernel<int><<<grid1d(n), 256, 0, ST>>>(sd, n, static_cast<const int*>(in), static_cast<int*>(out));
    else if (elem_bytes == 1) subset_nd_kernel<signed char><<<grid1d(n), 256, 0, ST>>>(sd, n, static_cast<const signed char*>(in), static_cast<signed char*>(out));
    else throw std::runtime_error("element size must be 1, 4 or 8 bytes");
    RRX_CATCH("rrx_subset_nd")
}

#define RRX_DEFINE_MISC(F, SFX) \
int rrx_increment_1scalar_by_1scalar##SFX(int ncol, int nlay, int ngpt, F* tau_inout, const F* tau_in, void* stream) \
{ RRX_TRY const size_t n = size_t(ncol)*nlay*ngpt; \
  inc_1scl_kernel<F><<<grid1d(n), 256, 0, ST>>>(n, tau_inout, tau_in); RRX_CATCH("rrx_increment_1scalar_by_1scalar") } \
int rrx_increment_2stream_by_2stream##SFX(int ncol, int nlay, int ngpt, F* tau_inout, F* ssa_inout, F* g_inout, const F* tau_in, const F* ssa_in, const F* g_in, void* stream) \
{ RRX_TRY const size_t n = size_t(ncol)*nlay*ngpt; \
  inc_2str_kernel<F><<<grid1d(n), 256, 0, ST>>>(n, rrx::Lim<F>::tiny()*F(3.), tau_inout, ssa_inout, g_inout, tau_in, ssa_in, g_in); RRX_CATCH("rrx_increment_2stream_by_2stream") } \
int rrx_inc_1scalar_by_1scalar_bybnd##SFX(int ncol, int nlay, int ngpt, F* tau_inout, const F* tau_in, int nbnd, const int* band_lims_gpoint, void* stream) \
{ RRX_TRY const size_t ncl = size_t(ncol)*nlay; \
  inc_1scl_bybnd_kernel<F><<<dim3(std::min(grid1d(ncl), 1024), ngpt), 256, 0, ST>>>(ncl, tau_inout, tau_in, nbnd, band_lims_gpoint); RRX_CATCH("rrx_inc_1scalar_by_1scalar_bybnd") } \
int rrx_inc_2stream_by_2stream_bybnd##SFX(int ncol, int nlay, int ngpt, F* tau_inout, F* ssa_inout, F* g_inout, const F* tau_in, const F* ssa_in, const F* g_in, int nbnd, const int* band_lims_gpoint, void* stream) \
{ RRX_TRY const size_t ncl = size_t(ncol)*nlay; \
  inc_2str_bybnd_kernel<F><<<dim3(std::min(grid1d(ncl), 1024), ngpt), 256, 0, ST>>>(ncl, rrx::Lim<F>::tiny()*F(3.), tau_inout, ssa_inout, g_inout, tau_in, ssa_in, g_in, nbnd, band_lims_gpoint); RRX_CATCH("rrx_inc_2stream_by_2stream_bybnd") } \
int rrx_delta_scale_2str_k##SFX(int ncol, int nlay, int ngpt, F* tau_inout, F* ssa_inout, F* g_inout, void* stream) \
{ RRX_TRY const size_t n = size_t(ncol)*nlay*ngpt; \
  delta_scale_kernel<F><<<grid1d(n), 256, 0, ST>>>(n, rrx::Lim<F>::tiny()*F(3.), tau_inout, ssa_inout, g_inout); RRX_CATCH("rrx_delta_scale_2str_k") } \
int rrx_sum_broadband##SFX(int ncol, int nlev, int ngpt, const F* gpt_flux, F* flux, void* stream) \
{ RRX_TRY const size_t n = size_t(ncol)*nlev; \
  sum_broadband_kernel<F><<<grid1d(n), 256, 0, ST>>>(n, ngpt, gpt_flux, flux); RRX_CATCH("rrx_sum_broadband") } \
int rrx_net_broadband_precalc##SFX(int ncol, int nlev, const F* flux_dn, const F* flux_up, F* flux_net, void* stream) \
{ RRX_TRY const size_t n = size_t(ncol)*nlev; \
  net_kernel<F><<<grid1d(n), 256, 0, ST>>>(n, flux_dn, flux_up, flux_net); RRX_CATCH("rrx_net_broadband_precalc") } \
int rrx_heating_rate##SFX(int ncol, int nlay, F g_over_cp, const F* flux_net, const F* plev, F* heating_rate, void* stream) \
{ RRX_TRY const size_t n = size_t(ncol)*nlay; \
  heating_rate_kernel<F><<<grid1d(n), 256, 0, ST>>>(n, ncol, g_over_cp, flux_net, plev, heating_rate); RRX_CATCH("rrx_heating_rate") } \
int rrx_sum_byband##SFX(int ncol, int nlev, int ngpt, int nbnd, const int* band_lims, const F* gpt_flux, F* bnd_flux, void* stream) \
{ RRX_TRY (void)ngpt; const size_t n = size_t(ncol)*nlev; \
  byband_kernel<F,false><<<dim3(std::min(grid1d(n), 1024), nbnd), 256, 0, ST>>>(n, band_lims, gpt_flux, (const F*)nullptr, bnd_flux); RRX_CATCH("rrx_sum_byband") } \
int rrx_net_byband_full##SFX(int ncol, int nlev, int ngpt, int nbnd, const int* band_lims, const F* gpt_flux_dn, const F* gpt_flux_up, F* bnd_flux_net, void* stream) \
{ RRX_TRY (void)ngpt; const size_t n = size_t(ncol)*nlev; \
  byband_kernel<F,true><<<dim3(std::min(grid1d(n), 1024), nbnd), 256, 0, ST>>>(n, band_lims, gpt_flux_dn, gpt_flux_up, bnd_flux_net); RRX_CATCH("rrx_net_byband_full") } \
int rrx_get_from_subset##SFX(int ncol, int nlay, int nbnd, int ncol_in, int col_s_in, int narr, F* const* var_full, const F* const* var_sub, void* stream) \
{ RRX_TRY if (narr < 1 || narr > 4) throw std::runtime_error("narr must be 1..4"); \
  if (col_s_in < 1 || col_s_in - 1 + ncol_in > ncol) throw std::runtime_error("column range outside the full array"); \
  PtrPack<F> p; for (int a=0; a<narr; ++a) { p.full[a] = var_full[a]; p.sub[a] = var_sub[a]; } \
  const size_t nrest = size_t(nlay)*nbnd; \
  get_from_subset_kernel<F><<<grid1d(size_t(ncol_in)*nrest), 256, 0, ST>>>(ncol, nrest, ncol_in, col_s_in, narr, p); RRX_CATCH("rrx_get_from_subset") } \
int rrx_fill_gases##SFX(int ncol, int nlay, int dim1, int dim2, int ngas, int igas, F* vmr_out, const F* vmr_in, F* col_gas, const F* col_dry, void* stream) \
{ RRX_TRY (void)ngas; fill_gases_kernel<F><<<grid1d(size_t(ncol)*nlay), 256, 0, ST>>>(ncol, nlay, dim1, dim2, igas, vmr_out, vmr_in, col_gas, col_dry); RRX_CATCH("rrx_fill_gases") } \
int rrx_fill_gases_all##SFX(int ncol, int nlay, int ngas, const F* const* vmr_in, const int* dim1, const int* dim2, F* col_gas, const F* col_dry, void* stream) \
{ RRX_TRY if (ngas < 0 || ngas > RRX_MAX_GASES) throw std::runtime_error("more gases than rrx_fill_gases_all takes"); \
  GasTable<F> gt; for (int i=0; i<ngas; ++i) { gt.vmr[i] = vmr_in[i]; gt.dim1[i] = dim1[i]; gt.dim2[i] = dim2[i]; } \
  fill_gases_all_kernel<F><<<grid1d(size_t(ncol)*nlay), 256, 0, ST>>>(ncol, nlay, ngas, gt, col_gas, col_dry); RRX_CATCH("rrx_fill_gases_all") } \
int rrx_get_col_dry##SFX(int ncol, int nlay, const F* vmr_h2o, const F* plev, F* col_dry, void* stream) \
{ RRX_TRY col_dry_kernel<F><<<grid1d(size_t(ncol)*nlay), 256, 0, ST>>>(ncol, nlay, vmr_h2o, plev, col_dry); RRX_CATCH("rrx_get_col_dry") } \
int rrx_expand_and_transpose##SFX(int ncol, int nbnd, const int* band_lims_gpt, const F* arr_in, F* arr_out, void* stream) \
{ RRX_TRY expand_and_transpose_kernel<F><<<dim3(std::min(grid1d(ncol), 1024), nbnd), 256, 0, ST>>>(ncol, nbnd, band_lims_gpt, arr_in, arr_out); RRX_CATCH("rrx_expand_and_transpose") } \
int rrx_spread_col##SFX(int ncol, int ngpt, F* toa_src, const F* solar_source, void* stream) \
{ RRX_TRY spread_col_kernel<F><<<dim3(std::min(ceil_div(ncol, 256), 64), ngpt), 256, 0, ST>>>(ncol, ngpt, toa_src, solar_source); RRX_CATCH("rrx_spread_col") } \
int rrx_scaling_to_subset##SFX(int ncol, int ngpt, F* toa_src, const F* tsi_scaling, void* stream) \
{ RRX_TRY scale_cols_kernel<F><<<dim3(std::min(ceil_div(ncol, 256), 64), ngpt), 256, 0, ST>>>(ncol, ngpt, toa_src, tsi_scaling); RRX_CATCH("rrx_scaling_to_subset") } \
int rrx_toa_source##SFX(int ncol, int ngpt, F* toa_src, const F* solar_source, const F* tsi_scaling, void* stream) \
{ RRX_TRY if (ncol <= 0 || ngpt <= 0) throw std::runtime_error("empty problem"); \
  if (tsi_scaling == nullptr) spread_col_kernel<F><<<dim3(std::min(ceil_div(ncol, 256), 64), ngpt), 256, 0, ST>>>(ncol, ngpt, toa_src, solar_source); \
  else toa_source_kernel<F><<<dim3(std::min(ceil_div(ncol, 256), 64), ngpt), 256, 0, ST>>>(ncol, ngpt, toa_src, solar_source, tsi_scaling); RRX_CATCH("rrx_toa_source") } \
int rrx_aerosol_optics##SFX(int ncol, int nlay, int nbnd, int nhum, int nphobic, int nphilic, const F* const* aermr, const int* aermr_per_column, \
        const F* rh, const F* plev, const F* rh_upper, const F* mext_phobic, const F* ssa_phobic, const F* g_phobic, \
        const F* mext_philic, const F* ssa_philic, const F* g_philic, F* tau, F* ssa, F* g, void* stream) \
{ RRX_TRY if (nphobic < 11 || nphilic < 5) throw std::runtime_error("aerosol tables need >= 11 hydrophobic and >= 5 hydrophilic species"); \
  if (nhum < 1) throw std::runtime_error("no humidity classes"); \
  AerosolSpecies<F> sp; for (int a=0; a<11; ++a) { if (!aermr[a]) throw std::runtime_error("missing aerosol mixing ratio"); sp.mmr[a] = aermr[a]; sp.per_column[a] = aermr_per_column ? aermr_per_column[a] : 1; } \
  aerosol_optics_kernel<F><<<grid1d(size_t(ncol)*nlay), 256, 0, ST>>>(ncol, nlay, nbnd, nhum, sp, rh, plev, rh_upper, \
      mext_phobic, ssa_phobic, g_phobic, mext_philic, ssa_philic, g_philic, tau, ssa, g); RRX_CATCH("rrx_aerosol_optics") } \
int rrx_cloud_optics_2str##SFX(int ncol, int nlay, int nbnd, int nsize_liq, int nsize_ice, \
        F radliq_lwr, F radliq_upr, F diamice_lwr, F diamice_upr, \
        const F* lut_extliq, const F* lut_ssaliq, const F* lut_asyliq, const F* lut_extice, const F* lut_ssaice, const F* lut_asyice, \
        const F* clwp, const F* ciwp, const F* reliq, const F* deice, F* tau, F* ssa, F* g, void* stream) \
{ RRX_TRY const size_t ncl = size_t(ncol)*nlay; \
  cloud_optics_kernel<F,true,false><<<grid1d(ncl), 256, 0, ST>>>(ncl, nbnd, nsize_liq, nsize_ice, \
      radliq_lwr, (radliq_upr - radliq_lwr)/(nsize_liq - F(1.)), diamice_lwr, (diamice_upr - diamice_lwr)/(nsize_ice - F(1.)), \
      lut_extliq, lut_ssaliq, lut_asyliq, lut_extice, lut_ssaice, lut_asyice, clwp, ciwp, reliq, deice, tau, ssa, g); RRX_CATCH("rrx_cloud_optics_2str") } \
int rrx_cloud_optics_2str_delta##SFX(int ncol, int nlay, int nbnd, int nsize_liq, int nsize_ice, \
        F radliq_lwr, F radliq_upr, F diamice_lwr, F diamice_upr, \
        const F* lut_extliq, const F* lut_ssaliq, const F* lut_asyliq, const F* lut_extice, const F* lut_ssaice, const F* lut_asyice, \
        const F* clwp, const F* ciwp, const F* reliq, const F* deice, F* tau, F* ssa, F* g, void* stream) \
{ RRX_TRY const size_t ncl = size_t(ncol)*nlay; \
  cloud_optics_kernel<F,true,true><<<grid1d(ncl), 256, 0, ST>>>(ncl, nbnd, nsize_liq, nsize_ice, \
      radliq_lwr, (radliq_upr - radliq_lwr)/(nsize_liq - F(1.)), diamice_lwr, (diamice_upr - diamice_lwr)/(nsize_ice - F(1.)), \
      lut_extliq, lut_ssaliq, lut_asyliq, lut_extice, lut_ssaice, lut_asyice, clwp, ciwp, reliq, deice, tau, ssa, g); RRX_CATCH("rrx_cloud_optics_2str_delta") } \
int rrx_cloud_optics_1scl##SFX(int ncol, int nlay, int nbnd, int nsize_liq, int nsize_ice, \
        F radliq_lwr, F radliq_upr, F diamice_lwr, F diamice_upr, \
        const F* lut_extliq, const F* lut_ssaliq, const F* lut_asyliq, const F* lut_extice, const F* lut_ssaice, const F* lut_asyice, \
        const F* clwp, const F* ciwp, const F* reliq, const F* deice, F* tau, void* stream) \
{ RRX_TRY const size_t ncl = size_t(ncol)*nlay; \
  cloud_optics_kernel<F,false,false><<<grid1d(ncl), 256, 0, ST>>>(ncl, nbnd, nsize_liq, nsize_ice, \
      radliq_lwr, (radliq_upr - radliq_lwr)/(nsize_liq - F(1.)), diamice_lwr, (diamice_upr - diamice_lwr)/(nsize_ice - F(1.)), \
      lut_extliq, lut_ssaliq, lut_asyliq, lut_extice, lut_ssaice, lut_asyice, clwp, ciwp, reliq, deice, tau, (F*)nullptr, (F*)nullptr); RRX_CATCH("rrx_cloud_optics_1scl") } \
int rrx_subset_cols##SFX(int ncol_full, int nrest, int col_s, int ncol_sub, const F* in, F* out, void* stream) \
{ RRX_TRY if (col_s < 1 || col_s - 1 + ncol_sub > ncol_full) throw std::runtime_error("column range outside the full array"); \
  subset_cols_kernel<F><<<grid1d(size_t(ncol_sub)*nrest), 256, 0, ST>>>(ncol_full, size_t(nrest), col_s, ncol_sub, in, out); RRX_CATCH("rrx_subset_cols") } \
int rrx_subset_lastdim##SFX(int n1, int col_s, int ncol_sub, const F* in, F* out, void* stream) \
{ RRX_TRY if (hipMemcpyAsync(out, in + size_t(col_s-1)*n1, size_t(n1)*ncol_sub*sizeof(F), hipMemcpyDeviceToDevice, ST) != hipSuccess) \
      throw std::runtime_error("hipMemcpyAsync failed"); RRX_CATCH("rrx_subset_lastdim") } \
int rrx_fill##SFX(unsigned long long n, F value, F* arr, void* stream) \
{ RRX_TRY fill_kernel<F><<<grid1d(n), 256, 0, ST>>>(n, value, arr); RRX_CATCH("rrx_fill") }

RRX_DEFINE_MISC(double, _f64)
RRX_DEFINE_MISC(float, _f32)
}
