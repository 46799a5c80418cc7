// Gas optics: interpolation, major+minor absorption, Rayleigh, combine, Planck source.
// Replaces Gas_optics_rrtmgp_kernels_cuda::* (/root/reference/src_kernels_cuda/gas_optics_rrtmgp_kernels_launchers.cu)
// and the kernels of /root/reference/src_kernels_cuda/gas_optics_rrtmgp_kernels.cu.
//
// MI355X design (DESIGN.md section "gas optics"): one thread per (column, layer) cell, lanes = 64 consecutive
// columns of one layer, each thread loops over the g-points. The per-cell interpolation state (jtemp, jpress,
// jeta, col_mix, fmajor, fminor of the current flavor) stays in registers, so it is read once per flavor change
// instead of once per g-point, and every (col,lay,gpt) output is written exactly once, coalesced over columns
// (512 B per wave-store). The reference instead launches gpt-fastest threads that write uncoalesced and adds
// major / minor-lower / minor-upper in three read-modify-write passes over tau.
// The k-distribution tables are gathered through L1/L2 (neighbouring columns hit the same lines); the per-chunk
// lists of minor contributors are built once per workgroup in LDS.
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include "rrx_common.h"
#include "rrx_hip.h"

namespace
{
#ifndef RRX_GW_NT
#define RRX_GW_NT 1       // the cell arrays are written once and far exceed the caches: non-temporal stores keep them from evicting the LUTs
#endif
#ifndef RRX_GW_ABL
#define RRX_GW_ABL 0      // ablation builds (tools/ab_extra.sh, tools/gw_timing.sh): 1 = set-up only, 2 = no staging, 3 = no g-point loop, 7 = no stores
#endif
// The same store addressed as uniform 64-bit base + 32-bit unsigned lane offset: the instruction's own scalar-base form, written as
// such (through C++ the compiler folds base and offset back into a 64-bit address per lane, one v_lshl_add_u64 per store). Counted by
// the hardware's vmcnt like any store; the compiler does not know of it, which only makes its own waits conservative.
template<typename F> __device__ __forceinline__ void stream_store_sbase(const char* sbase, const unsigned voff, const F v)
{
    static_assert(sizeof(F) == 4 || sizeof(F) == 8, "dword or dwordx2");
    if (RRX_GW_ABL == 7) { if (v == F(-12345.678)) *reinterpret_cast<F*>(const_cast<char*>(sbase) + voff) = v; return; }
#if RRX_GW_NT
    if constexpr (sizeof(F) == 4) asm volatile("global_store_dword %0, %1, %2 nt" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, %2 nt" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
#else
    if constexpr (sizeof(F) == 4) asm volatile("global_store_dword %0, %1, %2" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
#endif
}
template<typename F> __device__ __forceinline__ void stream_store(F* p, const F v)
{
    if (RRX_GW_ABL == 7) { if (v == F(-12345.678)) *p = v; return; }
#if RRX_GW_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

using namespace rrx;

constexpr int GCH = 16;          // g-points per register chunk
#ifndef RRX_GATHER_SHARES
#define RRX_GATHER_SHARES 4
#endif
constexpr int GSH = RRX_GATHER_SHARES;   // a handed-back workgroup is redone in this many shares of its g-point chunks (few entries: the launch lasts as long as one share)


// /root/reference/src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:317-395
template<typename F>
__global__ void __launch_bounds__(256)
interpolation_kernel(
        const int ncol, const int nlay, const int ngas, const int nflav, const int neta, const int npres, const int ntemp,
        const int* __restrict__ flavor, const F* __restrict__ press_ref_log, const F* __restrict__ temp_ref,
        const F press_ref_log_delta, const F temp_ref_min, const F temp_ref_delta, const F press_ref_trop_log,
        const F* __restrict__ vmr_ref, const F* __restrict__ play, const F* __restrict__ tlay,
        const F* __restrict__ col_gas,
        int* __restrict__ jtemp, F* __restrict__ fmajor, F* __restrict__ fminor, F* __restrict__ col_mix,
        Bool* __restrict__ tropo, int* __restrict__ jeta, int* __restrict__ jpress)
{
    const int icol = blockIdx.x*blockDim.x + threadIdx.x;
    const int ilay = blockIdx.y*blockDim.y + threadIdx.y;
    if (icol >= ncol || ilay >= nlay) return;

    const size_t ncl = size_t(ncol)*nlay;
    const size_t idx = icol + size_t(ilay)*ncol;
    const F tiny = Lim<F>::tiny();

    const F t = tlay[idx];
    int jt = int((t - (temp_ref_min - temp_ref_delta)) / temp_ref_delta);
    jt = min(ntemp-1, max(1, jt));
    jtemp[idx] = jt;
    const F ftemp = (t - temp_ref[jt-1]) / temp_ref_delta;

    const F lp = log(play[idx]);
    const F locpress = F(1.) + (lp - press_ref_log[0]) / press_ref_log_delta;
    const int jp = min(npres-1, max(1, int(locpress)));
    jpress[idx] = jp;
    const F fpress = locpress - F(jp);

    const bool in_tropo = lp > press_ref_trop_log;
    tropo[idx] = in_tropo;
    const int itropo = in_tropo ? 0 : 1;

    for (int iflav=0; iflav<nflav; ++iflav)
    {
        const int gas1 = flavor[2*iflav];
        const int gas2 = flavor[2*iflav+1];
        const size_t cell = idx + iflav*ncl;
        const F cg1 = col_gas[idx + gas1*ncl];
        const F cg2 = col_gas[idx + gas2*ncl];

        #pragma unroll
        for (int itemp=0; itemp<2; ++itemp)
        {
            const size_t vbase = itropo + size_t(jt+itemp-1) * (ngas+1) * 2;
            const F ratio_eta_half = vmr_ref[vbase + 2*gas1] / vmr_ref[vbase + 2*gas2];
            const F cmix = cg1 + ratio_eta_half * cg2;
            col_mix[itemp + 2*cell] = cmix;

            const F eta = (cmix > F(2.)*tiny) ? cg1 / cmix : F(0.5);
            const F loceta = eta * F(neta-1);
            jeta[itemp + 2*cell] = min(int(loceta)+1, neta-1);
            const F feta = loceta - trunc(loceta);          // = fmod(loceta, 1) exactly (loceta >= 0), without fmod's division loop
            const F ftemp_term = F(1-itemp) + F(2*itemp-1)*ftemp;

            const F f0 = (F(1.)-feta) * ftemp_term;
            const F f1 = feta * ftemp_term;
            F* fmi = &fminor[2*(itemp + 2*cell)];
            fmi[0] = f0; fmi[1] = f1;
            F* fma = &fmajor[4*(itemp + 2*cell)];
            fma[0] = (F(1.)-fpress) * f0;
            fma[1] = (F(1.)-fpress) * f1;
            fma[2] = fpress * f0;
            fma[3] = fpress * f1;
        }
    }
}


// The interpolation state of ONE cell, recomputed inside the consumers ("direct" entry points) instead of being written by
// interpolation_kernel and read back: per cell 3 words in (p, T, two column amounts per flavor) against 16 words per flavor
// out and in again (at C4: 3.5 GB written + 4.5 GB read per chain, and two launches of 0.78 ms). Same expressions, same
// order as interpolation_kernel above (/root/reference/src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:317-395), so the
// two paths agree bit for bit (tests/test_gpu_parity.py::test_direct_gas_optics_equals_interpolation_path).
template<typename F>
struct InterpArgs
{
    int ngas; const int* flavor; const F* press_ref_log; const F* temp_ref;
    F press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log; const F* vmr_ref;
    // all-sky: optical properties by band (nbnd, nlay, ncol) -- clouds, aerosols -- added to the gas optics where it is stored
    // (increment_*_by_*_bybnd of optical_props_kernels.cu:31-139 folded into the producer); cld_lims = band_lims_gpt (2, nbnd)
    const F* cld_tau = nullptr; const F* cld_ssa = nullptr; const F* cld_g = nullptr; const int* cld_lims = nullptr;
};

// the arithmetic of inc_2stream_by_2stream_bybnd for one g-point of a cell (rrx_misc.hip:inc_2str, same order of operations)
// (FAST: the two divisions as Newton reciprocals -- the windowed kernel's own rounding, like its single-scattering albedo)
template<typename F, bool FAST = false>
__device__ __forceinline__ void add_by_band_2str(F& tau1, F& ssa1, F& g1, const F tau2, const F ssa2, const F g2)
{
    const F eps = Lim<F>::tiny()*F(3.);
    const F tau12 = tau1 + tau2;
    const F tauscat12 = (tau1 * ssa1) + (tau2 * ssa2);
    if constexpr (FAST)
    {
        g1 = ((tau1 * ssa1 * g1) + (tau2 * ssa2 * g2)) * fast_rcp(max(tauscat12, eps));
        ssa1 = tauscat12 * fast_rcp(max(eps, tau12));
    }
    else
    {
        g1 = ((tau1 * ssa1 * g1) + (tau2 * ssa2 * g2)) / max(tauscat12, eps);
        ssa1 = tauscat12 / max(eps, tau12);
    }
    tau1 = tau12;
}

// The same combination where the first operand is a gas (asymmetry identically zero): the term tau1 ssa1 g1 is an exact zero and is
// not formed, and the cloud's products tau2 ssa2 and (tau2 ssa2) g2 stand alone -- loop-invariant where a chunk lies in one band.
// Bit for bit the result of add_by_band_2str(tau1, ssa1, g1 = 0, ...).
template<typename F, bool FAST = false>
__device__ __forceinline__ void add_cloud_to_gas_2str(F& tau1, F& ssa1, F& g1, const F tau2, const F ssa2, const F g2)
{
    const F eps = Lim<F>::tiny()*F(3.);
    const F cw = tau2 * ssa2, cwg = cw * g2;
    const F tau12 = tau1 + tau2;
    const F tauscat12 = (tau1 * ssa1) + cw;
    if constexpr (FAST)
    {
        g1 = cwg * fast_rcp(max(tauscat12, eps));
        ssa1 = tauscat12 * fast_rcp(max(eps, tau12));
    }
    else
    {
        g1 = cwg / max(tauscat12, eps);
        ssa1 = tauscat12 / max(eps, tau12);
    }
    tau1 = tau12;
}

template<typename F>
struct CellState { int jt, jp_raw, itropo; F ftemp, fpress; };

template<typename F>
__device__ __forceinline__ CellState<F> cell_state(const InterpArgs<F>& ia, const int npres, const int ntemp, const F p, const F t)
{
    CellState<F> c;
    int jt = int((t - (ia.temp_ref_min - ia.temp_ref_delta)) / ia.temp_ref_delta);
    jt = min(ntemp-1, max(1, jt));
    c.jt = jt;
    c.ftemp = (t - ia.temp_ref[jt-1]) / ia.temp_ref_delta;
    const F lp = log(p);
    const F locpress = F(1.) + (lp - ia.press_ref_log[0]) / ia.press_ref_log_delta;
    c.jp_raw = min(npres-1, max(1, int(locpress)));
    c.fpress = locpress - F(c.jp_raw);
    c.itropo = (lp > ia.press_ref_trop_log) ? 0 : 1;
    return c;
}

// flavor-dependent part for temperature node itemp (0: jt-1, 1: jt): col_mix, jeta, fminor[2], fmajor[4]
template<typename F>
__device__ __forceinline__ void flavor_state(const InterpArgs<F>& ia, const CellState<F>& c, const int neta, const int itemp,
        const int gas1, const int gas2, const F cg1, const F cg2, F& cmix, int& je, F (&fmi)[2], F (&fma)[4])
{
    const size_t vbase = c.itropo + size_t(c.jt+itemp-1) * (ia.ngas+1) * 2;
    const F ratio_eta_half = ia.vmr_ref[vbase + 2*gas1] / ia.vmr_ref[vbase + 2*gas2];
    cmix = cg1 + ratio_eta_half * cg2;
    const F eta = (cmix > F(2.)*Lim<F>::tiny()) ? cg1 / cmix : F(0.5);
    const F loceta = eta * F(neta-1);
    je = min(int(loceta)+1, neta-1);
    const F feta = loceta - trunc(loceta);                  // = fmod(loceta, 1) exactly (loceta >= 0), without fmod's division loop
    const F ftemp_term = F(1-itemp) + F(2*itemp-1)*c.ftemp;
    const F f0 = (F(1.)-feta) * ftemp_term;
    const F f1 = feta * ftemp_term;
    fmi[0] = f0; fmi[1] = f1;
    fma[0] = (F(1.)-c.fpress) * f0;
    fma[1] = (F(1.)-c.fpress) * f1;
    fma[2] = c.fpress * f0;
    fma[3] = c.fpress * f1;
}


// Per-workgroup tables in LDS, built once per workgroup:
//   gflav[r][ig]           flavor (0-based) of g-point ig in regime r (0 = lower, 1 = upper atmosphere)
//   lists[r][c] = { count, then per item {imnr, gpt_start, gpt_end, kminor_start-1-gpt_start, flavor} } : the minor
//   contributors overlapping 16-g-point chunk c, in ascending imnr (deterministic summation order, identical to the
//   reference's sequential loop over imnr).
// With these, the g-point loop has no dependent global loads for metadata.
constexpr int ITEM = 5;
struct MinorIndex
{
    const int* base; int stride_c; int stride_r;
    __device__ __forceinline__ int count(int r, int c) const { return base[r*stride_r + c*stride_c]; }
    __device__ __forceinline__ const int* item(int r, int c, int i) const { return base + r*stride_r + c*stride_c + 1 + ITEM*i; }
};

constexpr int MM = 8;        // ints per contributor in the LDS constant table
__device__ __forceinline__ int rfl(const int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ inline MinorIndex build_minor_index(
        int* lds, const int nchunk, const int nmax, const int* gflav, const int ngpt,
        const int* mmeta, const int nminorlower, const int nminorupper)
{
    MinorIndex mi{lds, 1 + ITEM*nmax, nchunk*(1 + ITEM*nmax)};
    const int tid = threadIdx.y*blockDim.x + threadIdx.x;
    const int nthr = blockDim.x*blockDim.y;
    for (int w = tid; w < 2*nchunk; w += nthr)
    {
        const int r = w / nchunk, c = w % nchunk;
        const int n = r == 0 ? nminorlower : nminorupper;
        int* out = lds + r*mi.stride_r + c*mi.stride_c;
        int cnt = 0;
        for (int i=0; i<n; ++i)
        {
            const int* m = mmeta + MM*(r*nmax + i);
            const int lo = m[4]-1, hi = m[5];                    // [lo, hi) zero-based
            if (lo < (c+1)*GCH && hi > c*GCH)
            {
                int* it = out + 1 + ITEM*cnt;
                it[0] = i; it[1] = lo; it[2] = hi; it[3] = m[6]-1 - lo; it[4] = gflav[r*ngpt + lo];
                ++cnt;
            }
        }
        out[0] = cnt;
    }
    return mi;
}

#ifndef RRX_GO_NPRE
#define RRX_GO_NPRE 1
#endif
#ifndef RRX_GO_NPRE32
#define RRX_GO_NPRE32 1
#endif
[[maybe_unused]] constexpr int NPRE_F64 = RRX_GO_NPRE, NPRE_F32 = RRX_GO_NPRE32;   // minor contributors requested in the first batch of a g-point group
constexpr int SL = 6;        // minor contributors of a chunk held in registers; further ones take a slower loop
struct Slots { int lo[SL], hi[SL], koff[SL], mf[SL]; };

// element at BYTE offset `boff` (32-bit, unsigned) from a wave-uniform base: lets the compiler use the
// scalar-base + 32-bit-vector-offset form of global_load instead of 64-bit per-lane address arithmetic
template<typename F>
__device__ __forceinline__ F ld(const F* __restrict__ base, const unsigned boff)
{
    return *reinterpret_cast<const F*>(reinterpret_cast<const char*>(base) + boff);
}


// The k tables have temperature as their fastest dimension, so the two temperature nodes of one (eta, pressure) corner
// are adjacent words: ld2 fetches both with one 2-word load (element-aligned only). The L1 serves a wave's gather at
// 4 lanes per clock whatever the width per lane (PMC: ~16 cache accesses per 64-lane load), so this halves the cost of
// every gather whose two temperature nodes share the eta index (je0 == je1; the other lanes issue the second node's
// loads under their own exec mask).
template<typename F> struct Pair { F x, y; };
template<typename F>
__device__ __forceinline__ Pair<F> ld2(const F* __restrict__ base, const unsigned boff)
{
    typedef F Vec2 __attribute__((ext_vector_type(2), aligned(sizeof(F))));
    const Vec2 v = *reinterpret_cast<const Vec2*>(reinterpret_cast<const char*>(base) + boff);
    return Pair<F>{v.x, v.y};
}


// MODE 0: tau += major + minor            (compute_tau_absorption, reference semantics: caller zeroes tau)
// MODE 1: tau/ssa/g = fused absorption + Rayleigh + combine   (SW gas optics in one pass)
// MODE 2: tau  = major + minor            (LW gas optics without the zero fill and the read-back of MODE 0)
// major : /root/reference/src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:398-443
// minor : :458-578      rayleigh : :674-718      combine : :721-746 with the CPU threshold (src/Gas_optics_rrtmgp.cpp:378)
// A minor interval uses the flavor of its first g-point, exactly as the reference kernel (:533).
// DIRECT: the interpolation state is computed in the kernel from (play, tlay, col_gas) -- InterpArgs -- and the arrays
// tropo / col_mix / fmajor / fminor / jeta / jtemp / jpress are not touched (may be null).
template<typename F, int MODE, bool DIRECT = false, bool CLD = false>
__global__ void __launch_bounds__(256, 2)
tau_absorption_kernel(
        const int ncol, const int nlay, const int ngpt, const int neta, const int npres, const int ntemp,
        const int nminorlower, const int nminorupper, const int idx_h2o,
        const int* __restrict__ gpoint_flavor,
        const F* __restrict__ kmajor, const F* __restrict__ kminor_lower, const F* __restrict__ kminor_upper,
        const int* __restrict__ minor_limits_gpt_lower, const int* __restrict__ minor_limits_gpt_upper,
        const Bool* __restrict__ minor_scales_with_density_lower, const Bool* __restrict__ minor_scales_with_density_upper,
        const Bool* __restrict__ scale_by_complement_lower, const Bool* __restrict__ scale_by_complement_upper,
        const int* __restrict__ idx_minor_lower, const int* __restrict__ idx_minor_upper,
        const int* __restrict__ idx_minor_scaling_lower, const int* __restrict__ idx_minor_scaling_upper,
        const int* __restrict__ kminor_start_lower, const int* __restrict__ kminor_start_upper,
        const Bool* __restrict__ tropo, const F* __restrict__ col_mix, const F* __restrict__ fmajor, const F* __restrict__ fminor,
        const F* __restrict__ play, const F* __restrict__ tlay, const F* __restrict__ col_gas, const F* __restrict__ col_dry,
        const int* __restrict__ jeta, const int* __restrict__ jtemp, const int* __restrict__ jpress,
        const F* __restrict__ krayl,
        F* __restrict__ tau, F* __restrict__ ssa, F* __restrict__ g, const InterpArgs<F> ia,
        const int* __restrict__ todo = nullptr, const int todo_gx = 1, const int todo_nblk = 1, const int todo_nz = 1, const int todo_geom = 0)
{
    // todo != null: this launch finishes the workgroups the windowed kernel handed back (gas_window_kernel): a 1-D grid, block b
    // takes over entry todo[1+b] = workgroup + part * todo_nblk of the (todo_gx x . x todo_nz) grid -- a part is a share of the
    // g-point chunks (gas_window_kernel's grid.z); blocks beyond the count todo[0] have nothing to do.
    // (Round 2 measured four gather workgroups per entry at C4: 0.44 -> 0.38 ms for SW, 0.27 -> 0.31 ms for LW -- the per-workgroup
    //  set-up dominated. With the set-up paid once per resident workgroup, below, an entry is redone in GSH shares of its chunks:
    //  a launch with a handful of entries lasts as long as one share, not as long as one whole workgroup's 256 g-points.)
    // Round 3: the grid is capped (gather_grid) and a workgroup takes entries one after the other (the first by its index, the next
    // from a counter in the list's header: todo[-1] for this kernel, todo[-2] for planck_fraction_kernel): the usual launch
    // -- nothing handed back -- starts 2 048 workgroups that leave at once instead of one per windowed workgroup (0.12 -> 0.03 ms
    // at C4), and a workgroup that does take entries builds its index tables once for all of them.
    extern __shared__ int lds_int[];
    const int nchunk = (ngpt + GCH - 1) / GCH;
    const int n_entries = (todo != nullptr) ? todo[0]*GSH : 1;          // work items: (entry, share of its chunks)
    if (todo != nullptr && int(blockIdx.x) >= n_entries) return;
    const int nmax = max(nminorlower, nminorupper);
    int* gflav = lds_int;                                   // [2][ngpt]
    int* gchg = lds_int + 2*ngpt;                           // [ngpt] 1 where the flavor of either regime changes
    int* lists = lds_int + 3*ngpt;
    // per-contributor constants {idx_minor, scales_with_density, idx_minor_scaling, scale_by_complement} of both regimes:
    // read from LDS in the chunk set-up, so that the per-cell scalings need no chain of dependent global loads
    int* mmeta = lists + 2*nchunk*(1 + ITEM*nmax);           // [2][nmax][MM]: + {gpt_start, gpt_end, kminor_start}
    {
        const int tid = threadIdx.y*blockDim.x + threadIdx.x;
        for (int w = tid; w < 2*ngpt; w += blockDim.x*blockDim.y)
            gflav[(w & 1)*ngpt + (w >> 1)] = gpoint_flavor[w] - 1;
        for (int w = tid; w < ngpt; w += blockDim.x*blockDim.y)
            gchg[w] = (w > 0 && (gpoint_flavor[2*w] != gpoint_flavor[2*w-2] || gpoint_flavor[2*w+1] != gpoint_flavor[2*w-1])) ? 1 : 0;
        for (int w = tid; w < nminorlower; w += blockDim.x*blockDim.y)
        {
            int* m = mmeta + MM*w;
            m[0] = idx_minor_lower[w]; m[1] = minor_scales_with_density_lower[w] ? 1 : 0;
            m[2] = idx_minor_scaling_lower[w]; m[3] = scale_by_complement_lower[w] ? 1 : 0;
            m[4] = minor_limits_gpt_lower[2*w]; m[5] = minor_limits_gpt_lower[2*w+1]; m[6] = kminor_start_lower[w];
        }
        for (int w = tid; w < nminorupper; w += blockDim.x*blockDim.y)
        {
            int* m = mmeta + MM*(nmax + w);
            m[0] = idx_minor_upper[w]; m[1] = minor_scales_with_density_upper[w] ? 1 : 0;
            m[2] = idx_minor_scaling_upper[w]; m[3] = scale_by_complement_upper[w] ? 1 : 0;
            m[4] = minor_limits_gpt_upper[2*w]; m[5] = minor_limits_gpt_upper[2*w+1]; m[6] = kminor_start_upper[w];
        }
    }
    __syncthreads();                  // the chunk lists below are built from the LDS copy of the interval limits
    const MinorIndex mi = build_minor_index(lists, nchunk, nmax, gflav, ngpt, mmeta, nminorlower, nminorupper);
    __syncthreads();

    __shared__ int s_next;
    for (int ientry = (todo != nullptr) ? int(blockIdx.x) : 0; ientry < n_entries; )
    {
    // the next entry of this workgroup: taken from a counter (entries differ in cost, a fixed stride left the last workgroups alone)
    const int ientry_now = ientry;
    if (todo != nullptr)
    {
        __syncthreads();
        if (threadIdx.x == 0 && threadIdx.y == 0) s_next = int(gridDim.x) + atomicAdd(const_cast<int*>(todo) - 1, 1);
        __syncthreads();
        ientry = s_next;
    }
    else ientry = n_entries;
    int blk_x = blockIdx.x, blk_y = blockIdx.y;
    int c_lo = 0, c_hi = nchunk;
    if (todo != nullptr)
    {
        const int entry = todo[1 + ientry_now / GSH], share = ientry_now % GSH;
        const int part = entry / todo_nblk, blk = entry % todo_nblk;
        blk_x = blk % todo_gx; blk_y = blk / todo_gx;
        const int per = (nchunk + todo_nz - 1) / todo_nz;
        if (part < todo_nz) { c_lo = part*per; c_hi = min(nchunk, c_lo + per); }      // (part == todo_nz: the whole range, see gas_window_kernel)
        const int q = (c_hi - c_lo + GSH - 1) / GSH;
        c_lo += share*q; c_hi = min(c_hi, c_lo + q);
        if (c_lo >= c_hi) continue;
    }
    // (todo_geom 1: the handed-back workgroup was 256 columns of one layer, see gas_window_geometry)
    const int icol = todo_geom ? (blk_x*4 + int(threadIdx.y))*64 + int(threadIdx.x) : blk_x*blockDim.x + threadIdx.x;
    const int ilay = todo_geom ? blk_y : blk_y*blockDim.y + threadIdx.y;
    // (control flow stays uniform across the workgroup here: every thread reaches the loop latch and its barriers; a per-thread
    //  `continue` past them was a divergent barrier -- ADVICE r03)
    const bool cell_ok = icol < ncol && ilay < nlay;
    if (cell_ok)
    {

    const size_t ncl = size_t(ncol)*nlay;
    const size_t idx = icol + size_t(ilay)*ncol;
    const F pl = play[idx], tl = tlay[idx];
    CellState<F> cs;
    if constexpr (DIRECT) cs = cell_state<F>(ia, npres, ntemp, pl, tl);
    else { cs.itropo = tropo[idx] ? 0 : 1; cs.jt = jtemp[idx]; cs.jp_raw = jpress[idx]; cs.ftemp = F(0.); cs.fpress = F(0.); }
    const int itropo = cs.itropo;
    const int jt = cs.jt;
    const int jp = cs.jp_raw + itropo;
    const int s_eta = ntemp, s_prs = ntemp*neta;
    const size_t s_gpt = size_t(ntemp)*neta*(npres+1);
    const int tn = ntemp*neta;
    constexpr unsigned SZ = sizeof(F);

    const F cdry0 = col_gas[idx];                         // col_gas(:,:,0) = col_dry
    const F ch2o = col_gas[idx + size_t(idx_h2o)*ncl];
    F ray_fac = F(0.);
    if constexpr (MODE == 1) ray_fac = ch2o + col_dry[idx];

    // One pass per regime (lower / upper atmosphere): inside a pass every active lane is in regime `itr`, so everything
    // that depends on the regime and the g-point only -- flavor, minor-contributor lists, table bases, loop bounds -- is
    // wave-uniform. readfirstlane (first ACTIVE lane) puts those values into SGPRs: conditions and table addresses
    // become scalar, inactive contributors cost a scalar branch. A wavefront whose columns are all in one regime (the
    // rule: pressure is nearly constant along a level) skips the other pass; one that straddles the tropopause runs both.
    for (int itr=0; itr<2; ++itr)
    {
    if (itropo != itr) continue;
    // /root/reference/src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:505-529; both column amounts are requested at once
    // (the scaling gas defaults to index 0 = col_dry where there is none: loaded, not used)
    auto minor_scaling = [&](const int imnr) -> F
    {
        const int* m = mmeta + MM*(itr*nmax + imnr);
        const int imn = rfl(m[0]), swd = rfl(m[1]), ims = rfl(m[2]), sbc = rfl(m[3]);
        F scaling = col_gas[idx + size_t(imn)*ncl];
        const F cscal = col_gas[idx + size_t(max(ims, 0))*ncl];
        if (swd)
        {
            scaling *= F(0.01) * pl / tl;
            if (ims > 0)
            {
                const F vmr_fact = F(1.) / cdry0;
                const F dry_fact = F(1.) / (F(1.) + ch2o * vmr_fact);
                const F x = cscal * vmr_fact * dry_fact;
                scaling *= sbc ? (F(1.) - x) : x;
            }
        }
        return scaling;
    };

    // interpolation state of the current flavor; LUT offsets are BYTE offsets within one g-point slab / table row
    int cur_flav = -1;
    bool same_eta = false;                       // je0 == je1: the jt node sits one word after the jt-1 node
    F fm0=0, fm1=0, fm2=0, fm3=0, fm4=0, fm5=0, fm6=0, fm7=0, cm0=0, cm1=0, fn0=0, fn1=0, fn2=0, fn3=0;
    unsigned b00=0, b01=0, b10=0, b11=0;       // kmajor: (jt-1 | jt) x (jp-1 | jp), lower eta node
    unsigned q0a=0, q0b=0, q1a=0, q1b=0;       // kminor / krayl
    const unsigned beta = unsigned(s_eta)*SZ;

    // interpolation state of a flavor: from the arrays, or (DIRECT) computed here
    auto flavor_of_cell = [&](const int iflav, F (&cm)[2], int (&je)[2], F (&fn)[4], F (&fm)[8])
    {
        if constexpr (DIRECT)
        {
            const int gas1 = rfl(ia.flavor[2*iflav]), gas2 = rfl(ia.flavor[2*iflav+1]);      // iflav is wave-uniform
            const F cg1 = col_gas[idx + size_t(gas1)*ncl], cg2 = col_gas[idx + size_t(gas2)*ncl];
            #pragma unroll
            for (int itemp=0; itemp<2; ++itemp)
            {
                F fmi[2], fma[4];
                flavor_state<F>(ia, cs, neta, itemp, gas1, gas2, cg1, cg2, cm[itemp], je[itemp], fmi, fma);
                fn[2*itemp] = fmi[0]; fn[2*itemp+1] = fmi[1];
                fm[4*itemp] = fma[0]; fm[4*itemp+1] = fma[1]; fm[4*itemp+2] = fma[2]; fm[4*itemp+3] = fma[3];
            }
        }
        else
        {
            const size_t cell = idx + size_t(iflav)*ncl;
            #pragma unroll
            for (int i=0; i<8; ++i) fm[i] = fmajor[8*cell + i];
            cm[0] = col_mix[2*cell]; cm[1] = col_mix[2*cell+1];
            je[0] = jeta[2*cell]; je[1] = jeta[2*cell+1];
            #pragma unroll
            for (int i=0; i<4; ++i) fn[i] = fminor[4*cell + i];
        }
    };

    auto load_flavor = [&](const int iflav)
    {
        cur_flav = iflav;
        F cm[2], fn[4], fm[8]; int je[2];
        flavor_of_cell(iflav, cm, je, fn, fm);
        fm0=fm[0]; fm1=fm[1]; fm2=fm[2]; fm3=fm[3]; fm4=fm[4]; fm5=fm[5]; fm6=fm[6]; fm7=fm[7];
        cm0 = cm[0]; cm1 = cm[1];
        const int je0 = je[0], je1 = je[1];
        fn0=fn[0]; fn1=fn[1]; fn2=fn[2]; fn3=fn[3];
        same_eta = (je0 == je1);
        b00 = unsigned((jt-1) + (je0-1)*s_eta + (jp-1)*s_prs)*SZ;  b01 = b00 + unsigned(s_prs)*SZ;
        b10 = unsigned( jt    + (je1-1)*s_eta + (jp-1)*s_prs)*SZ;  b11 = b10 + unsigned(s_prs)*SZ;
        q0a = unsigned((jt-1) + (je0-1)*ntemp)*SZ; q0b = q0a + unsigned(ntemp)*SZ;
        q1a = unsigned( jt    + (je1-1)*ntemp)*SZ; q1b = q1a + unsigned(ntemp)*SZ;
    };

    // minor absorption of a contributor whose flavor is not the band's (does not occur in rrtmgp-data; kept exact)
    auto minor_other_flavor = [&](const int mflav, const F* km) -> F
    {
        F cm[2], fn[4], fm[8]; int je[2];
        flavor_of_cell(mflav, cm, je, fn, fm);
        const int j0 = je[0], j1 = je[1];
        return fn[0]*km[(jt-1) + (j0-1)*ntemp] + fn[1]*km[(jt-1) + j0*ntemp]
             + fn[2]*km[ jt    + (j1-1)*ntemp] + fn[3]*km[ jt    + j1*ntemp];
    };

    // G g-points at a time: all LUT gathers of the group are issued before the first use, so one memory round trip is
    // paid per group instead of per g-point (a g-point-at-a-time loop is latency-bound: 78 % of wave cycles in s_waitcnt).
    // Batch shape, measured per form at C4 (alternating builds on one box, tools/ab_build.sh; differences are 2-4 %):
    //   fp64 SW fused form: 2 g-points, no contributor in the first batch  -> 157 VGPRs, 3 waves per SIMD (5.68 vs 5.80 ms)
    //   fp64 LW forms:      4 g-points + 2 contributors in the first batch -> 2 waves per SIMD          (4.30 vs 4.43 ms)
    //   fp32:               4 g-points + 1 contributor, 3 waves per SIMD
    // RRX_GO_G / RRX_GO_NPRE / RRX_GO_NPRE32 override all of them for A/B builds.
#ifdef RRX_GO_G
    constexpr int G = RRX_GO_G;
    constexpr int NPRE = (sizeof(F) == 8) ? NPRE_F64 : NPRE_F32;
#else
    constexpr int G = (sizeof(F) == 8 && MODE == 1) ? 2 : 4;
    constexpr int NPRE = (sizeof(F) == 8) ? ((MODE == 1) ? 0 : 2) : 1;
#endif
    [[maybe_unused]] int cb = 0, cb_have = -1;   // CLD: band (0-based) of the g-point being stored; g-points ascend within a pass
    [[maybe_unused]] F c_tau = F(0.), c_ssa = F(0.), c_g = F(0.);
    auto gpoint_group = [&](const int ig0, const int gend, const int c, const int n, const Slots& sl, const F (&sc)[SL])
    {
        int igs[G];
        #pragma unroll
        for (int u=0; u<G; ++u) igs[u] = min(ig0 + u, gend-1);

        F kv[G][8];
        #pragma unroll
        for (int u=0; u<G; ++u)
        {
            const F* k = kmajor + size_t(igs[u])*s_gpt;            // wave-uniform base
            const Pair<F> p0 = ld2(k, b00), p1 = ld2(k, b00 + beta), p2 = ld2(k, b01), p3 = ld2(k, b01 + beta);
            kv[u][0] = p0.x; kv[u][1] = p1.x; kv[u][2] = p2.x; kv[u][3] = p3.x;
            kv[u][4] = p0.y; kv[u][5] = p1.y; kv[u][6] = p2.y; kv[u][7] = p3.y;
        }
        if (!same_eta)
        {
            #pragma unroll
            for (int u=0; u<G; ++u)
            {
                const F* k = kmajor + size_t(igs[u])*s_gpt;
                kv[u][4] = ld(k, b10); kv[u][5] = ld(k, b10 + beta); kv[u][6] = ld(k, b11); kv[u][7] = ld(k, b11 + beta);
            }
        }
        F rv[G][4];
        if constexpr (MODE == 1)
        {
            #pragma unroll
            for (int u=0; u<G; ++u)
            {
                const F* kr = krayl + size_t(itr)*tn*ngpt + size_t(igs[u])*tn;
                const Pair<F> r0 = ld2(kr, q0a), r1 = ld2(kr, q0b);
                rv[u][0] = r0.x; rv[u][1] = r1.x; rv[u][2] = r0.y; rv[u][3] = r1.y;
            }
            if (!same_eta)
            {
                #pragma unroll
                for (int u=0; u<G; ++u)
                {
                    const F* kr = krayl + size_t(itr)*tn*ngpt + size_t(igs[u])*tn;
                    rv[u][2] = ld(kr, q1a); rv[u][3] = ld(kr, q1b);
                }
            }
        }
        F told[G];
        if constexpr (MODE == 0)
        {
            #pragma unroll
            for (int u=0; u<G; ++u) told[u] = tau[idx + size_t(igs[u])*ncl];
        }

        const F* kmin = itr == 0 ? kminor_lower : kminor_upper;
        auto minor_active = [&](const int i) -> bool { return i < n && ig0 < sl.hi[i] && ig0 + G > sl.lo[i]; };
        auto minor_load = [&](const int i, F (&mv)[G][4])
        {
            if constexpr (sizeof(F) == 4)
            {
                // fp32: one g-point at a time (the batched form below costs this precision its third wave per SIMD:
                // 163 -> 191 VGPRs, SW 4.4 -> 5.1 ms)
                #pragma unroll
                for (int u=0; u<G; ++u)
                {
                    const int kg = min(max(igs[u], sl.lo[i]), sl.hi[i]-1);      // clamped: always a valid table row
                    const F* km = kmin + size_t(kg + sl.koff[i])*tn;
                    if (sl.mf[i] == cur_flav)
                    {
                        const Pair<F> m0 = ld2(km, q0a), m1 = ld2(km, q0b);
                        mv[u][0] = m0.x; mv[u][1] = m1.x; mv[u][2] = m0.y; mv[u][3] = m1.y;
                        if (!same_eta) { mv[u][2] = ld(km, q1a); mv[u][3] = ld(km, q1b); }
                    }
                    else { mv[u][0] = minor_other_flavor(sl.mf[i], km); mv[u][1] = mv[u][2] = mv[u][3] = F(0.); }
                }
            }
            else if (sl.mf[i] == cur_flav)                                 // wave-uniform
            {
                // fp64: straight-line per contributor -- the 2-word loads of all G g-points first, the separate jt-node
                // loads of lanes whose two temperatures sit in different eta intervals in ONE divergent block afterwards
                // (a branch inside the g-point loop splits the batch into G dependent pieces): LW 4.55 -> 4.35 ms
                const F* km[G];
                #pragma unroll
                for (int u=0; u<G; ++u)
                {
                    const int kg = min(max(igs[u], sl.lo[i]), sl.hi[i]-1);
                    km[u] = kmin + size_t(kg + sl.koff[i])*tn;
                    const Pair<F> m0 = ld2(km[u], q0a), m1 = ld2(km[u], q0b);
                    mv[u][0] = m0.x; mv[u][1] = m1.x; mv[u][2] = m0.y; mv[u][3] = m1.y;
                }
                if (!same_eta)
                {
                    #pragma unroll
                    for (int u=0; u<G; ++u) { mv[u][2] = ld(km[u], q1a); mv[u][3] = ld(km[u], q1b); }
                }
            }
            else
            {
                #pragma unroll
                for (int u=0; u<G; ++u)
                {
                    const int kg = min(max(igs[u], sl.lo[i]), sl.hi[i]-1);
                    mv[u][0] = minor_other_flavor(sl.mf[i], kmin + size_t(kg + sl.koff[i])*tn);
                    mv[u][1] = mv[u][2] = mv[u][3] = F(0.);
                }
            }
        };

        // the first NPRE contributors of the chunk are requested together with the major / Rayleigh words: one memory
        // round trip for the whole group where a g-point has at most NPRE contributors (the rule in the upper atmosphere)
        F mvp[NPRE > 0 ? NPRE : 1][G][4];
        #pragma unroll
        for (int i=0; i<NPRE; ++i)
            if (minor_active(i)) minor_load(i, mvp[i]);

        F t[G];
        #pragma unroll
        for (int u=0; u<G; ++u)
            t[u] = cm0 * (fm0*kv[u][0] + fm1*kv[u][1] + fm2*kv[u][2] + fm3*kv[u][3])
                 + cm1 * (fm4*kv[u][4] + fm5*kv[u][5] + fm6*kv[u][6] + fm7*kv[u][7]);

        auto minor_accum = [&](const int i, const F (&mv)[G][4])
        {
            // branch-free over the g-points of the group (rows outside the contributor's interval were loaded from its
            // nearest valid row and are discarded by the select): no control flow splits the group's instruction stream
            const bool own = (sl.mf[i] == cur_flav);
            #pragma unroll
            for (int u=0; u<G; ++u)
            {
                const F kk = own ? fn0*mv[u][0] + fn1*mv[u][1] + fn2*mv[u][2] + fn3*mv[u][3] : mv[u][0];
                const F tn_ = t[u] + kk * sc[i];
                t[u] = (igs[u] >= sl.lo[i] && igs[u] < sl.hi[i]) ? tn_ : t[u];
            }
        };
        #pragma unroll
        for (int i=0; i<SL; ++i)                          // ascending contributor index: the reference's summation order
        {
            if (minor_active(i))
            {
                if (i < NPRE) minor_accum(i, mvp[i]);
                else { F mv[G][4]; minor_load(i, mv); minor_accum(i, mv); }
            }
        }
        for (int i=SL; i<n; ++i)                          // more than SL contributors in one chunk: rare
        {
            const int* itp = mi.item(itr, c, i);
            const int it[5] = {rfl(itp[0]), rfl(itp[1]), rfl(itp[2]), rfl(itp[3]), rfl(itp[4])};
            #pragma unroll
            for (int u=0; u<G; ++u)
                if (igs[u] >= it[1] && igs[u] < it[2])
                {
                    const F* km = kmin + size_t(igs[u] + it[3])*tn;
                    const F kk = (it[4] == cur_flav) ? fn0*ld(km, q0a) + fn1*ld(km, q0b) + fn2*ld(km, q1a) + fn3*ld(km, q1b)
                                                     : minor_other_flavor(it[4], km);
                    t[u] += kk * minor_scaling(it[0]);
                }
        }

        #pragma unroll
        for (int u=0; u<G; ++u)
        {
            if (ig0 + u < gend)
            {
                const size_t o = idx + size_t(ig0 + u)*ncl;
                if constexpr (MODE == 0)
                {
                    tau[o] = told[u] + t[u];
                }
                else if constexpr (MODE == 2)
                {
                    if constexpr (CLD)
                    {
                        while (ig0 + u + 1 > ia.cld_lims[2*cb+1]) ++cb;
                        if (cb != cb_have) { cb_have = cb; c_tau = ia.cld_tau[idx + size_t(cb)*ncl]; }
                        stream_store(tau + o, t[u] + c_tau);
                    }
                    else stream_store(tau + o, t[u]);
                }
                else
                {
                    const F ray = ray_fac * (fn0*rv[u][0] + fn1*rv[u][1] + fn2*rv[u][2] + fn3*rv[u][3]);
                    F tt = t[u] + ray;
                    F ww = (tt > F(2.)*Lim<F>::eps()) ? ray / tt : F(0.);
                    if constexpr (CLD)
                    {
                        while (ig0 + u + 1 > ia.cld_lims[2*cb+1]) ++cb;
                        if (cb != cb_have)
                        {
                            cb_have = cb;
                            const size_t b = idx + size_t(cb)*ncl;
                            c_tau = ia.cld_tau[b]; c_ssa = ia.cld_ssa[b]; c_g = ia.cld_g[b];
                        }
                        F gg = F(0.);
                        add_by_band_2str(tt, ww, gg, c_tau, c_ssa, c_g);
                        stream_store(tau + o, tt); stream_store(ssa + o, ww); stream_store(g + o, gg);
                    }
                    else
                    {
                        stream_store(tau + o, tt);
                        stream_store(ssa + o, ww);
                        if (g != nullptr) stream_store(g + o, F(0.));
                    }
                }
            }
        }
    };

    for (int c=c_lo; c<c_hi; ++c)
    {
        const int c0 = c*GCH;
        const int n = rfl(mi.count(itr, c));
        const int gend = min(c0 + GCH, ngpt);

        // this chunk's minor contributors: parameters and per-cell scaling in registers
        Slots sl; F sc[SL];
        #pragma unroll
        for (int i=0; i<SL; ++i)
        {
            const int* it = mi.item(itr, c, min(i, max(n-1, 0)));
            sl.lo[i] = rfl(it[1]); sl.hi[i] = rfl(it[2]); sl.koff[i] = rfl(it[3]); sl.mf[i] = rfl(it[4]);
            sc[i] = F(0.);
            if (i < n) sc[i] = minor_scaling(rfl(it[0]));
        }

        for (int ig0=c0; ig0<gend; )
        {
            // a group never straddles a flavor change (of either regime, so that group bounds stay wave-uniform)
            const int iflav = rfl(gflav[itr*ngpt + ig0]);
            if (iflav != cur_flav) load_flavor(iflav);
            int ge = min(ig0 + G, gend);
            #pragma unroll
            for (int u=G-1; u>=1; --u)
                if (ig0 + u < gend && rfl(gchg[ig0 + u])) ge = ig0 + u;
            gpoint_group(ig0, ge, c, n, sl, sc);
            ig0 = ge;
        }
    }
    }   // regime passes
    }   // cell_ok
    }   // entries
}


// /root/reference/src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:674-718 (standalone launcher parity)
template<typename F>
__global__ void __launch_bounds__(256)
tau_rayleigh_kernel(
        const int ncol, const int nlay, const int ngpt, const int neta, const int ntemp,
        const int* __restrict__ gpoint_flavor, const F* __restrict__ krayl,
        const int idx_h2o, const F* __restrict__ col_dry, const F* __restrict__ col_gas,
        const F* __restrict__ fminor, const int* __restrict__ jeta, const Bool* __restrict__ tropo,
        const int* __restrict__ jtemp, F* __restrict__ tau_rayleigh)
{
    const int icol = blockIdx.x*blockDim.x + threadIdx.x;
    const int ilay = blockIdx.y*blockDim.y + threadIdx.y;
    if (icol >= ncol || ilay >= nlay) return;
    const size_t ncl = size_t(ncol)*nlay;
    const size_t idx = icol + size_t(ilay)*ncol;
    const int itropo = tropo[idx] ? 0 : 1;
    const int jt = jtemp[idx];
    const F fac = col_gas[idx + size_t(idx_h2o)*ncl] + col_dry[idx];
    int cur_flav = -1;
    F f[4]; int j0 = 1, j1 = 1;
    for (int ig=0; ig<ngpt; ++ig)
    {
        const int iflav = gpoint_flavor[itropo + 2*ig] - 1;
        if (iflav != cur_flav)
        {
            cur_flav = iflav;
            const size_t cell = idx + iflav*ncl;
            #pragma unroll
            for (int i=0; i<4; ++i) f[i] = fminor[4*cell + i];
            j0 = jeta[2*cell]; j1 = jeta[2*cell+1];
        }
        const F* k = krayl + size_t(itropo)*ntemp*neta*ngpt + size_t(ig)*ntemp*neta;
        const F kloc = f[0] * k[(jt-1) + (j0-1)*ntemp] + f[1] * k[(jt-1) + j0*ntemp] +
                       f[2] * k[ jt    + (j1-1)*ntemp] + f[3] * k[ jt    + j1*ntemp];
        tau_rayleigh[idx + size_t(ig)*ncl] = kloc * fac;
    }
}


template<typename F>
__global__ void combine_kernel(const size_t n, const F* __restrict__ tau_abs, const F* __restrict__ tau_ray,
        F* __restrict__ tau, F* __restrict__ ssa, F* __restrict__ g)
{
    for (size_t i = size_t(blockIdx.x)*blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x)*blockDim.x)
    {
        const F r = tau_ray[i];
        const F t = tau_abs[i] + r;
        tau[i] = t;
        ssa[i] = (t > F(2.)*Lim<F>::eps()) ? r / t : F(0.);
        g[i] = F(0.);
    }
}


// /root/reference/src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:2-13
template<typename F>
__device__ __forceinline__ F interp1d(const F val, const F offset, const F delta, const int len, const F* __restrict__ table)
{
    const F val0 = (val - offset)/delta;
    const F frac = val0 - int(val0);
    const int idx = min(len-1, max(1, int(val0)+1));
    return table[idx-1] + frac * (table[idx] - table[idx-1]);
}
// the same in two steps, for one value looked up in many tables (the band Planck functions of all bands): position once, then per table
template<typename F> struct Interp1dPos { F frac; int idx; };
template<typename F>
__device__ __forceinline__ Interp1dPos<F> interp1d_pos(const F val, const F offset, const F delta, const int len)
{
    const F val0 = (val - offset)/delta;
    return Interp1dPos<F>{val0 - int(val0), min(len-1, max(1, int(val0)+1))};
}
template<typename F>
__device__ __forceinline__ F interp1d_at(const Interp1dPos<F> p, const F* __restrict__ table)
{
    return table[p.idx-1] + p.frac * (table[p.idx] - table[p.idx-1]);
}

template<typename F>
struct CellInterp
{
    F fm[8]; int je[2]; int jt, jp;
    CellState<F> cs;                 // DIRECT form only
    __device__ __forceinline__ void load(const size_t cell, const F* __restrict__ fmajor, const int* __restrict__ jeta)
    {
        #pragma unroll
        for (int i=0; i<8; ++i) fm[i] = fmajor[8*cell + i];
        je[0] = jeta[2*cell]; je[1] = jeta[2*cell+1];
    }
    // the same state computed from the column amounts of the flavor's two gases (see flavor_state)
    __device__ __forceinline__ void load_direct(const InterpArgs<F>& ia, const int neta, const int iflav,
                                                const F* __restrict__ col_gas, const size_t idx, const size_t ncl)
    {
        const int gas1 = ia.flavor[2*iflav], gas2 = ia.flavor[2*iflav+1];
        const F cg1 = col_gas[idx + size_t(gas1)*ncl], cg2 = col_gas[idx + size_t(gas2)*ncl];
        #pragma unroll
        for (int itemp=0; itemp<2; ++itemp)
        {
            F cmix, fmi[2], fma[4];
            flavor_state<F>(ia, cs, neta, itemp, gas1, gas2, cg1, cg2, cmix, je[itemp], fmi, fma);
            fm[4*itemp] = fma[0]; fm[4*itemp+1] = fma[1]; fm[4*itemp+2] = fma[2]; fm[4*itemp+3] = fma[3];
        }
    }
    __device__ __forceinline__ F pfrac(const F* __restrict__ p, const size_t s_eta, const size_t s_prs) const
    {
        return (fm[0] * p[(jt-1) + (je[0]-1)*s_eta + (jp-1)*s_prs]
              + fm[1] * p[(jt-1) +  je[0]   *s_eta + (jp-1)*s_prs]
              + fm[2] * p[(jt-1) + (je[0]-1)*s_eta +  jp   *s_prs]
              + fm[3] * p[(jt-1) +  je[0]   *s_eta +  jp   *s_prs])
             + (fm[4] * p[ jt    + (je[1]-1)*s_eta + (jp-1)*s_prs]
              + fm[5] * p[ jt    +  je[1]   *s_eta + (jp-1)*s_prs]
              + fm[6] * p[ jt    + (je[1]-1)*s_eta +  jp   *s_prs]
              + fm[7] * p[ jt    +  je[1]   *s_eta +  jp   *s_prs]);
    }
};

#ifndef RRX_PLANCK_MINWAVES
#define RRX_PLANCK_MINWAVES 1
#endif
#ifndef RRX_PLANCK_PL
#define RRX_PLANCK_PL 4
#endif
constexpr int PL = RRX_PLANCK_PL; // layers per Planck workgroup (64 columns x PL layers): 4 = three workgroups per CU whose gather and store phases overlap (8: one; measured 4.07 -> 3.90 ms fp64, 3.36 -> 2.53 ms fp32)

// /root/reference/src_kernels_cuda/gas_optics_rrtmgp_kernels.cu:196-314
// The reference recomputes the Planck fraction of the layer below for every level source (16 LUT gathers per
// cell). Here a workgroup of 64 columns x 8 layers exchanges the fractions through LDS in chunks of 16 g-points,
// so only the first layer of each workgroup recomputes its neighbour (9 gathers per cell on average).
template<typename F, bool DIRECT = false>
__global__ void __launch_bounds__(64*PL, RRX_PLANCK_MINWAVES)
planck_source_kernel(
        const int ncol, const int nlay, const int ngpt, const int neta, const int npres, const int ntemp, const int nPlanckTemp,
        const F* __restrict__ tlay, const F* __restrict__ tlev, const F* __restrict__ tsfc, const int sfc_lay,
        const F* __restrict__ fmajor, const int* __restrict__ jeta, const Bool* __restrict__ tropo,
        const int* __restrict__ jtemp, const int* __restrict__ jpress,
        const int* __restrict__ gpoint_bands, const F* __restrict__ pfracin,
        const F temp_ref_min, const F totplnk_delta, const F* __restrict__ totplnk,
        const int* __restrict__ gpoint_flavor,
        F* __restrict__ sfc_src, F* __restrict__ lay_src, F* __restrict__ lev_src, F* __restrict__ sfc_src_jac, const int share_on,
        const F* __restrict__ play, const F* __restrict__ col_gas, const InterpArgs<F> ia)
{
    extern __shared__ double lds_raw[];
    F* pf = reinterpret_cast<F*>(lds_raw);            // [GCH][PL+1][64]; slot 0 = layer below the workgroup
    int* gflav = reinterpret_cast<int*>(pf + GCH*(PL+1)*64);   // [2][ngpt] flavor (0-based) per regime and g-point
    // per-wavefront staging area of the shared-cell path below: [2 sets][4 corners][16 g-points] temperature pairs
    typedef F Vec2 __attribute__((ext_vector_type(2)));
    typedef F Vec2u __attribute__((ext_vector_type(2), aligned(sizeof(F))));
    Vec2* stg = reinterpret_cast<Vec2*>(gflav + ((2*ngpt + 3) & ~3)) + size_t(threadIdx.y)*2*4*GCH;
    for (int w = threadIdx.y*64 + threadIdx.x; w < 2*ngpt; w += 64*PL)
        gflav[(w & 1)*ngpt + (w >> 1)] = gpoint_flavor[w] - 1;
    __syncthreads();

    const int tx = threadIdx.x, ly = threadIdx.y;
    const int icol_raw = blockIdx.x*64 + tx;
    const int ilay_raw = blockIdx.y*PL + ly;
    const bool active = icol_raw < ncol && ilay_raw < nlay;
    const int icol = min(icol_raw, ncol-1);
    const int ilay = min(ilay_raw, nlay-1);

    const size_t ncl = size_t(ncol)*nlay;
    const size_t ncv = size_t(ncol)*(nlay+1);
    const size_t idx = icol + size_t(ilay)*ncol;
    const size_t s_eta = ntemp, s_prs = size_t(ntemp)*neta, s_gpt = size_t(ntemp)*neta*(npres+1);
    const F delta_Tsurf = F(1.);

    CellInterp<F> own, prev;
    int itropo;
    if constexpr (DIRECT)
    {
        own.cs = cell_state<F>(ia, npres, ntemp, play[idx], tlay[idx]);
        itropo = own.cs.itropo; own.jt = own.cs.jt; own.jp = own.cs.jp_raw + itropo;
    }
    else
    {
        itropo = tropo[idx] ? 0 : 1;
        own.jt = jtemp[idx]; own.jp = jpress[idx] + itropo;
    }
    const bool has_prev = ilay > 0;
    // The layer below the workgroup's first one (its fractions enter lev_src of that first layer) is shared out: every
    // wavefront computes GCH/PL of its g-points per chunk, instead of wavefront 0 doing a second full layer while the
    // others wait at the barrier (ablation: the fraction phase was 1.8 of 3.55 ms and did not overlap with the rest).
    const int lay0 = blockIdx.y*PL;
    const bool halo = lay0 > 0;                        // workgroup-uniform
    const size_t idx_h = size_t(icol) + size_t(max(lay0-1, 0))*ncol;
    int itropo_m1 = 0;
    if (halo)
    {
        if constexpr (DIRECT)
        {
            prev.cs = cell_state<F>(ia, npres, ntemp, play[idx_h], tlay[idx_h]);
            itropo_m1 = prev.cs.itropo; prev.jt = prev.cs.jt; prev.jp = prev.cs.jp_raw + itropo_m1;
        }
        else
        {
            itropo_m1 = tropo[idx_h] ? 0 : 1;
            prev.jt = jtemp[idx_h]; prev.jp = jpress[idx_h] + itropo_m1;
        }
    }
    const F t_lay = tlay[idx], t_lev = tlev[idx];
    const bool is_last = ilay == nlay-1;
    const bool is_sfc = ilay == sfc_lay-1;
    const F t_levp = tlev[idx + ncol];
    const F t_sfc = tsfc[icol];

    int cur_flav = -1, cur_flav_m1 = -1, cur_bnd = -1;
    F b_lay = 0, b_lev = 0, b_levp = 0, b_sfc = 0, b_sfc2 = 0;

    // Planck fractions of up to PG g-points of one cell: all 8*PG gathers are issued before the first use
    // batch of the per-lane (not shared-cell) gathers: small on purpose -- at 2 (fp64: 164 VGPRs) / 1 (fp32: 118) a third
    // workgroup fits each CU, whose phases interleave with the others': 3.26 -> 2.63 ms (fp32 1.96 -> 1.74), and still
    // faster than 4 when every wavefront takes this path (3.35 against 3.63 ms)
    constexpr int PG = (sizeof(F) == 8) ? 2 : 1;
    auto fractions = [&](CellInterp<F>& ci, int& cur, const size_t cell_idx, const int itr, const int ig_first, const int gend, const int slot,
                         const int ig0)
    {
        // one regime per wavefront <=> the flavor of a g-point is the same in all lanes and this loop runs convergent
        const bool one_regime = share_on && __all(itr == __builtin_amdgcn_readfirstlane(itr));
        for (int ig=ig_first; ig<gend; )
        {
            const int fl = gflav[itr*ngpt + ig];
            if (fl != cur)
            {
                cur = fl;
                if constexpr (DIRECT) ci.load_direct(ia, neta, fl, col_gas, cell_idx, ncl);
                else ci.load(cell_idx + size_t(fl)*ncl, fmajor, jeta);
            }

            // Shared-cell path: when the 64 columns of the wavefront sit in the same LUT cell for this flavor (the rule at
            // one level of an LES domain), the 8 corners x up to 16 g-points of the band are fetched ONCE per wavefront --
            // lane = (corner pair, g-point), one or two 2-word loads -- and every lane combines them with its own weights
            // through broadcast LDS reads: 1-2 wave-loads per band instead of 4 per g-point. Same words, same sums.
            if (one_regime)
            {
                const int jt0 = __builtin_amdgcn_readfirstlane(ci.jt), jp0 = __builtin_amdgcn_readfirstlane(ci.jp);
                const int e0 = __builtin_amdgcn_readfirstlane(ci.je[0]), e1 = __builtin_amdgcn_readfirstlane(ci.je[1]);
                if (__all(ci.jt == jt0 && ci.jp == jp0 && ci.je[0] == e0 && ci.je[1] == e1))
                {
                    int gr = ig + 1;
                    while (gr < gend && gflav[itr*ngpt + gr] == fl) ++gr;
                    const int ng = gr - ig;                                   // <= GCH
                    const int c = tx >> 4, gi = tx & 15;
                    const size_t oc = size_t(jt0-1) + size_t(e0-1 + (c & 1))*s_eta + size_t(jp0-1 + (c >> 1))*s_prs;
                    Vec2 qa = Vec2{F(0.), F(0.)}, qb = qa;
                    if (gi < ng)
                    {
                        const F* src = pfracin + size_t(ig + gi)*s_gpt + oc;
                        qa = *reinterpret_cast<const Vec2u*>(src);
                        if (e0 != e1) qb = *reinterpret_cast<const Vec2u*>(src + (e1 - e0)*ptrdiff_t(s_eta));
                    }
                    __builtin_amdgcn_wave_barrier();
                    stg[c*GCH + gi] = qa;
                    stg[(4 + c)*GCH + gi] = qb;
                    __builtin_amdgcn_wave_barrier();
                    #pragma unroll 4
                    for (int g=0; g<ng; ++g)
                    {
                        const Vec2 q0 = stg[g], q1 = stg[GCH + g], q2 = stg[2*GCH + g], q3 = stg[3*GCH + g];
                        F v4 = q0.y, v5 = q1.y, v6 = q2.y, v7 = q3.y;
                        if (e0 != e1) { v4 = stg[4*GCH + g].y; v5 = stg[5*GCH + g].y; v6 = stg[6*GCH + g].y; v7 = stg[7*GCH + g].y; }
                        pf[((ig + g - ig0)*(PL+1) + slot)*64 + tx] =
                            (ci.fm[0]*q0.x + ci.fm[1]*q1.x + ci.fm[2]*q2.x + ci.fm[3]*q3.x)
                          + (ci.fm[4]*v4 + ci.fm[5]*v5 + ci.fm[6]*v6 + ci.fm[7]*v7);
                    }
                    ig = gr;
                    continue;
                }
            }
            int ge = min(ig + PG, gend);
            #pragma unroll
            for (int u=PG-1; u>=1; --u)
                if (ig + u < gend && gflav[itr*ngpt + ig + u] != fl) ge = ig + u;
            F v[PG][8];
            #pragma unroll
            for (int u=0; u<PG; ++u)
            {
                const F* p = pfracin + size_t(min(ig + u, ge-1))*s_gpt;
                // both temperature nodes of a corner with one 2-word load (see ld2); node jt separately where je differs
                const size_t o00 = (ci.jt-1) + (ci.je[0]-1)*s_eta + (ci.jp-1)*s_prs;
                const Pair<F> p0 = ld2(p + o00, 0u), p1 = ld2(p + o00 + s_eta, 0u);
                const Pair<F> p2 = ld2(p + o00 + s_prs, 0u), p3 = ld2(p + o00 + s_prs + s_eta, 0u);
                v[u][0] = p0.x; v[u][1] = p1.x; v[u][2] = p2.x; v[u][3] = p3.x;
                v[u][4] = p0.y; v[u][5] = p1.y; v[u][6] = p2.y; v[u][7] = p3.y;
            }
            if (ci.je[0] != ci.je[1])
            {
                #pragma unroll
                for (int u=0; u<PG; ++u)
                {
                    const F* p = pfracin + size_t(min(ig + u, ge-1))*s_gpt;
                    v[u][4] = p[ ci.jt    + (ci.je[1]-1)*s_eta + (ci.jp-1)*s_prs]; v[u][5] = p[ ci.jt    + ci.je[1]*s_eta + (ci.jp-1)*s_prs];
                    v[u][6] = p[ ci.jt    + (ci.je[1]-1)*s_eta +  ci.jp   *s_prs]; v[u][7] = p[ ci.jt    + ci.je[1]*s_eta +  ci.jp   *s_prs];
                }
            }
            #pragma unroll
            for (int u=0; u<PG; ++u)
                if (ig + u < ge)
                    pf[((ig + u - ig0)*(PL+1) + slot)*64 + tx] =
                        (ci.fm[0]*v[u][0] + ci.fm[1]*v[u][1] + ci.fm[2]*v[u][2] + ci.fm[3]*v[u][3])
                      + (ci.fm[4]*v[u][4] + ci.fm[5]*v[u][5] + ci.fm[6]*v[u][6] + ci.fm[7]*v[u][7]);
            ig = ge;
        }
    };

    for (int c0=0; c0<ngpt; c0+=GCH)
    {
        const int gend = min(c0 + GCH, ngpt);
        fractions(own, cur_flav, idx, itropo, c0, gend, ly+1, c0);
        if (halo)
        {
            const int q = (gend - c0 + PL - 1) / PL;
            const int hb = c0 + ly*q, he = min(hb + q, gend);
            if (hb < he) fractions(prev, cur_flav_m1, idx_h, itropo_m1, hb, he, 0, c0);
        }
        __syncthreads();

        for (int ig=c0; ig<gend; ++ig)
        {
            const int u = ig - c0;
            const int ibnd = gpoint_bands[ig] - 1;
            if (ibnd != cur_bnd)
            {
                cur_bnd = ibnd;
                const F* tp = totplnk + size_t(ibnd)*nPlanckTemp;
                b_lay = interp1d(t_lay, temp_ref_min, totplnk_delta, nPlanckTemp, tp);
                b_lev = interp1d(t_lev, temp_ref_min, totplnk_delta, nPlanckTemp, tp);
                if (is_last) b_levp = interp1d(t_levp, temp_ref_min, totplnk_delta, nPlanckTemp, tp);
                if (is_sfc)
                {
                    b_sfc  = interp1d(t_sfc              , temp_ref_min, totplnk_delta, nPlanckTemp, tp);
                    b_sfc2 = interp1d(t_sfc + delta_Tsurf, temp_ref_min, totplnk_delta, nPlanckTemp, tp);
                }
            }
            if (active)
            {
                const F pfrac = pf[(u*(PL+1) + ly+1)*64 + tx];
                stream_store(lay_src + idx + size_t(ig)*ncl, pfrac * b_lay);
                F lev_val = pfrac * b_lev;
                if (has_prev) lev_val = sqrt(pfrac * pf[(u*(PL+1) + ly)*64 + tx]) * b_lev;
                stream_store(lev_src + idx + size_t(ig)*ncv, lev_val);
                if (is_last) lev_src[idx + ncol + size_t(ig)*ncv] = pfrac * b_levp;
                if (is_sfc)
                {
                    sfc_src    [icol + size_t(ig)*ncol] = pfrac * b_sfc;
                    sfc_src_jac[icol + size_t(ig)*ncol] = pfrac * (b_sfc2 - b_sfc);
                }
            }
        }
        __syncthreads();
    }
}



// "Planck-lite": the Planck fractions pfrac(col,lay,gpt) (8-point interpolation in planck_frac, as CellInterp::pfrac),
// the band-integrated Planck functions B(tlay)(col,lay,bnd) and B(tlev)(col,lev,bnd), and the surface terms -- everything
// Planck_source_kernel (gas_optics_rrtmgp_kernels.cu:196-314) computes EXCEPT the two products lay_source = pfrac*B_lay and
// lev_source = sqrt(pfrac*pfrac')*B_lev, which the broadband LW solver forms itself (rrx_lw_solver_noscat_fractions) or
// rrx_planck_sources_from_fractions materialises for anybody else. One (col,lay,gpt) array written instead of two, no
// neighbour-layer exchange. One thread per cell, lanes = 64 consecutive columns, interpolation state computed in place.
template<typename F>
__global__ void __launch_bounds__(256)
planck_fraction_kernel(
        const int ncol, const int nlay, const int ngpt, const int neta, const int npres, const int ntemp, const int nPlanckTemp,
        const F* __restrict__ play, const F* __restrict__ tlay, const F* __restrict__ tlev, const F* __restrict__ tsfc, const int sfc_lay,
        const F* __restrict__ col_gas, const InterpArgs<F> ia,
        const int* __restrict__ gpoint_bands, const F* __restrict__ pfracin,
        const F totplnk_delta, const F* __restrict__ totplnk, const int* __restrict__ gpoint_flavor,
        F* __restrict__ pfrac_out, F* __restrict__ blay_out, F* __restrict__ blev_out,
        F* __restrict__ sfc_src, F* __restrict__ sfc_src_jac, const int* __restrict__ todo = nullptr, const int todo_gx = 1,
        const int todo_nblk = 1, const int todo_nz = 1, const int todo_geom = 0)
{
    const int n_entries = (todo != nullptr) ? todo[0]*GSH : 1;  // todo: see tau_absorption_kernel (capped grid, work items taken from a counter)
    if (todo != nullptr && int(blockIdx.x) >= n_entries) return;
    extern __shared__ int lds_gflav[];                       // [2][ngpt] flavor (0-based) per regime and g-point
    for (int w = threadIdx.y*64 + threadIdx.x; w < 2*ngpt; w += 64*blockDim.y)
        lds_gflav[(w & 1)*ngpt + (w >> 1)] = gpoint_flavor[w] - 1;
    __syncthreads();

    __shared__ int s_next;
    for (int ientry = (todo != nullptr) ? int(blockIdx.x) : 0; ientry < n_entries; )
    {
    const int ientry_now = ientry;                           // (next entry from a counter of its own: todo[-2])
    if (todo != nullptr)
    {
        __syncthreads();
        if (threadIdx.x == 0 && threadIdx.y == 0) s_next = int(gridDim.x) + atomicAdd(const_cast<int*>(todo) - 2, 1);
        __syncthreads();
        ientry = s_next;
    }
    else ientry = n_entries;
    int blk_x = blockIdx.x, blk_y = blockIdx.y;
    int g_lo = 0, g_hi = ngpt;
    if (todo != nullptr)
    {
        const int entry = todo[1 + ientry_now / GSH], share = ientry_now % GSH;
        const int part = entry / todo_nblk, blk = entry % todo_nblk;
        blk_x = blk % todo_gx; blk_y = blk / todo_gx;
        const int nchunk = (ngpt + GCH - 1) / GCH, per = (nchunk + todo_nz - 1) / todo_nz;
        int c_lo = 0, c_hi = nchunk;
        if (part < todo_nz) { c_lo = part*per; c_hi = min(nchunk, c_lo + per); }      // (part == todo_nz: the whole range)
        const int q = (c_hi - c_lo + GSH - 1) / GSH;
        c_lo += share*q; c_hi = min(c_hi, c_lo + q);
        if (c_lo >= c_hi) continue;
        g_lo = min(c_lo*GCH, ngpt); g_hi = min(c_hi*GCH, ngpt);
    }
    const int icol = todo_geom ? (blk_x*4 + int(threadIdx.y))*64 + int(threadIdx.x) : blk_x*64 + threadIdx.x;
    const int ilay = todo_geom ? blk_y : blk_y*blockDim.y + threadIdx.y;
    // (control flow stays uniform across the workgroup here: every thread reaches the loop latch and its barriers; a per-thread
    //  `continue` past them was a divergent barrier -- ADVICE r03)
    const bool cell_ok = icol < ncol && ilay < nlay;
    if (cell_ok)
    {
    const size_t ncl = size_t(ncol)*nlay;
    const size_t ncv = size_t(ncol)*(nlay+1);
    const size_t idx = icol + size_t(ilay)*ncol;
    const unsigned s_eta = ntemp, s_prs = unsigned(ntemp)*neta;
    const size_t s_gpt = size_t(ntemp)*neta*(npres+1);
    constexpr unsigned SZ = sizeof(F);

    CellInterp<F> ci;
    ci.cs = cell_state<F>(ia, npres, ntemp, play[idx], tlay[idx]);
    const int itropo = ci.cs.itropo;
    ci.jt = ci.cs.jt; ci.jp = ci.cs.jp_raw + itropo;
    const F t_lay = tlay[idx], t_lev = tlev[idx];
    const bool is_last = ilay == nlay-1;
    const bool is_sfc = ilay == sfc_lay-1;
    const F t_levp = tlev[idx + ncol];
    const F t_sfc = tsfc[icol];

    int cur_flav = -1, cur_bnd = -1;
    F b_sfc = 0, b_sfc2 = 0;
    unsigned b0 = 0, b1 = 0;           // byte offsets of the (jt-1 | jt) pairs at (je0, jp-1) and (je1, jp-1)
    bool same_eta = false;
    constexpr int PG = 4;              // g-points whose gathers are in flight together

    for (int ig=g_lo; ig<g_hi; )
    {
        const int fl = lds_gflav[itropo*ngpt + ig];
        if (fl != cur_flav)
        {
            cur_flav = fl;
            ci.load_direct(ia, neta, fl, col_gas, idx, ncl);
            b0 = unsigned((ci.jt-1) + (ci.je[0]-1)*s_eta + (ci.jp-1)*s_prs)*SZ;
            b1 = unsigned( ci.jt    + (ci.je[1]-1)*s_eta + (ci.jp-1)*s_prs)*SZ;
            same_eta = (ci.je[0] == ci.je[1]);
        }
        int ge = min(ig + PG, g_hi);
        #pragma unroll
        for (int u=PG-1; u>=1; --u)
            if (ig + u < g_hi && lds_gflav[itropo*ngpt + ig + u] != fl) ge = ig + u;

        F v[PG][8];
        #pragma unroll
        for (int u=0; u<PG; ++u)
        {
            const F* p = pfracin + size_t(min(ig + u, ge-1))*s_gpt;
            const Pair<F> p0 = ld2(p, b0), p1 = ld2(p, b0 + s_eta*SZ), p2 = ld2(p, b0 + s_prs*SZ), p3 = ld2(p, b0 + (s_prs + s_eta)*SZ);
            v[u][0] = p0.x; v[u][1] = p1.x; v[u][2] = p2.x; v[u][3] = p3.x;
            v[u][4] = p0.y; v[u][5] = p1.y; v[u][6] = p2.y; v[u][7] = p3.y;
        }
        if (!same_eta)
        {
            #pragma unroll
            for (int u=0; u<PG; ++u)
            {
                const F* p = pfracin + size_t(min(ig + u, ge-1))*s_gpt;
                v[u][4] = ld(p, b1); v[u][5] = ld(p, b1 + s_eta*SZ); v[u][6] = ld(p, b1 + s_prs*SZ); v[u][7] = ld(p, b1 + (s_prs + s_eta)*SZ);
            }
        }
        #pragma unroll
        for (int u=0; u<PG; ++u)
        {
            const int g = ig + u;
            if (g < ge)
            {
                const F pfrac = (ci.fm[0]*v[u][0] + ci.fm[1]*v[u][1] + ci.fm[2]*v[u][2] + ci.fm[3]*v[u][3])
                              + (ci.fm[4]*v[u][4] + ci.fm[5]*v[u][5] + ci.fm[6]*v[u][6] + ci.fm[7]*v[u][7]);
                stream_store(pfrac_out + idx + size_t(g)*ncl, pfrac);
                const int ibnd = gpoint_bands[g] - 1;
                if (ibnd != cur_bnd)
                {
                    cur_bnd = ibnd;
                    const F* tp = totplnk + size_t(ibnd)*nPlanckTemp;
                    blay_out[idx + size_t(ibnd)*ncl] = interp1d(t_lay, ia.temp_ref_min, totplnk_delta, nPlanckTemp, tp);
                    blev_out[idx + size_t(ibnd)*ncv] = interp1d(t_lev, ia.temp_ref_min, totplnk_delta, nPlanckTemp, tp);
                    if (is_last) blev_out[idx + ncol + size_t(ibnd)*ncv] = interp1d(t_levp, ia.temp_ref_min, totplnk_delta, nPlanckTemp, tp);
                    if (is_sfc)
                    {
                        b_sfc  = interp1d(t_sfc        , ia.temp_ref_min, totplnk_delta, nPlanckTemp, tp);
                        b_sfc2 = interp1d(t_sfc + F(1.), ia.temp_ref_min, totplnk_delta, nPlanckTemp, tp);
                    }
                }
                if (is_sfc)
                {
                    sfc_src    [icol + size_t(g)*ncol] = pfrac * b_sfc;
                    sfc_src_jac[icol + size_t(g)*ncol] = pfrac * (b_sfc2 - b_sfc);
                }
            }
        }
        ig = ge;
    }
    }   // cell_ok
    }   // entries
}


// =====================================================================================================================
// Windowed gas optics. The gather kernel above is bound by the vector L1: every (cell, g-point) pulls 64 B of kmajor plus
// 32 B per minor contributor (and Rayleigh) through a 64 B/clk pipe, although the 64 columns of a wavefront sit in a handful
// of LUT cells. Here a workgroup (64 columns x 4 layers) stages, per 16-g-point chunk, the BOX of LUT nodes its cells use --
// NPW pressures x NEW etas x NTW temperature pairs, 12 KB of kmajor -- in LDS with a few coalesced loads, and every cell then
// reads its corners from LDS (256 B/clk, equal addresses broadcast): 32 x less traffic through L1. Same expressions in the
// same order as the gather kernel, so the same bits. A workgroup whose cells do not fit the box (columns in both regimes,
// a pressure / temperature / eta spread beyond the box, a chunk with a flavor change or more than NCW contributors) writes
// its id to a todo list and leaves; the gather kernel is launched behind on exactly those workgroups.
// PF: the Planck fractions ride along (planck_frac has kmajor's layout: same box, same corner weights) together with the band
// Planck functions and the surface terms -- the whole "Planck-lite" output of planck_fraction_kernel.
#ifndef RRX_GW_PAIR
#define RRX_GW_PAIR 1     // 1: g-points of a chunk go in pairs where the chunk allows it. Round 2: SW stage 3.81 -> 3.69 ms alone, but its
                          // registers collided with the batched register staging (3.81 -> 3.37 ms), so it was off. Round 3: with the boxes
                          // staged by LDS-DMA the registers are free: SW stage 2.50 -> 2.40 ms (same box), on.
#endif
#ifndef RRX_GW_LDSDMA
#define RRX_GW_LDSDMA 1   // boxes staged by LDS-DMA (global_load_lds_dwordx4) instead of through registers
#endif
#ifndef RRX_GW_FAST_BYBAND
#define RRX_GW_FAST_BYBAND 1  // all-sky SW form: the two divisions of the by-band combination as Newton reciprocals
#endif
#ifndef RRX_GW_BANDCHUNKS
#define RRX_GW_BANDCHUNKS 1   // chunks end where the flavor or the contributor set changes (0: every 16 g-points, the cut of rounds 1-3; A/B runs)
#endif
#ifndef RRX_GW_SPARSE
#define RRX_GW_SPARSE 1   // only the nodes the workgroup's cells reach are staged (their extent in pressure, eta and temperature), through registers
#endif
constexpr int NPW = 4, NEW = 4, NTW = 3;
constexpr int WBOX = NPW*NEW*NTW;            // pair-nodes per g-point: kmajor, planck_frac
constexpr int MBOX = NEW*NTW;                // pair-nodes per g-point: one minor contributor, Rayleigh
constexpr int NCW = 6;                       // minor contributors of a chunk with a staged window
constexpr int NXW = 12;                      // ... and how many a chunk may have at all: those beyond NCW are added in a pass of their own behind the g-point loop

// Parts (grid.z) the chunk loop of the windowed kernel is shared out over: 1 when the (column, layer) workgroups alone fill the
// chip once (three resident per CU), else 2 or 4
inline int gas_window_parts(const int nblk, const int nchunk)
{
    static const int forced = std::getenv("RRX_GW_PARTS") ? std::atoi(std::getenv("RRX_GW_PARTS")) : 0;      // (A/B runs)
    if (forced > 0) return std::min(forced, nchunk);
    // (round 4, RRX_GW_PARTS sweep at 2 048 and 4 096 columns: a split pays only while the workgroups do not fill the 768 resident
    //  places once -- every part repeats the set-up of its workgroup, a tenth of its life; 1 120 workgroups: 0.43 ms in one part or two)
    int nz = 1;
    while (nblk*nz < 768 && nz < 4 && 2*nz <= nchunk) nz *= 2;
    return nz;
}

// Workgroup shape of the windowed kernel: 256 cells that share LUT boxes. 64 columns x 4 layers (geom 0) put a regime change or a
// jump of the binary-species parameter BETWEEN the layers of one workgroup (at C4: 512 of 8 960 workgroups handed back, 0.8 ms of
// gather kernels per step); 256 columns x 1 layer (geom 1) have no vertical neighbours to disagree with and at C4 every workgroup
// fits its boxes. The wide shape is taken when the columns fill it; RRX_GW_GEOM=0/1 overrides (A/B runs).
inline int gas_window_geometry(const int ncol)
{
    if (const char* e = std::getenv("RRX_GW_GEOM")) return std::atoi(e) ? 1 : 0;
    return ncol >= 192 ? 1 : 0;
}
// grid of the gather kernels behind a windowed launch: at most this many workgroups walk over the todo list
inline dim3 gather_grid(const int entries_)
{
    const int entries = entries_*GSH;
    static const int cap = std::getenv("RRX_GATHER_GRID") ? std::max(1, std::atoi(std::getenv("RRX_GATHER_GRID"))) : 512;   // (A/B runs; 512 = the two workgroups per CU the kernels fit: the entries are taken from a counter anyway, and the usual launch -- nothing handed back -- costs 3 instead of 13 us)
    return dim3(std::min(entries, cap));
}
inline dim3 gas_window_grid(const int geom, const int ncol, const int nlay)
{
    return geom ? dim3(ceil_div(ncol, 256), nlay) : dim3(ceil_div(ncol, 64), ceil_div(nlay, 4));
}

// RRX_GW_STATS=1 (read at every launch, so a host program can switch it on for one solve): after a windowed launch, wait for it,
// print how many workgroups were handed back to the gather kernel and why, and add them to the calling thread's totals
// (rrx_gas_window_stats). Diagnostic only: it synchronises the stream.
#ifndef RRX_GW_TIMING
#define RRX_GW_TIMING 0   // diagnostic build (tools/gw_timing.sh): wavefront 0 of every workgroup adds the clocks it spends per phase to g_gw_clk, printed with RRX_GW_STATS
#endif
#if RRX_GW_TIMING
__device__ unsigned long long g_gw_clk[8];
#define RRX_GW_T(k) { const unsigned long long t_ = __builtin_readcyclecounter(); gw_acc[k] += t_ - gw_t; gw_t = t_; }
#else
#define RRX_GW_T(k)
#endif
thread_local long long g_gw_handed = 0, g_gw_total = 0;
inline void gas_window_stats(const char* what, const int* todo, const int nblk, hipStream_t st)
{
    if (std::getenv("RRX_GW_STATS") == nullptr) return;
    int h[9];
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(h, todo - 8, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
    g_gw_handed += h[8]; g_gw_total += nblk;
    std::fprintf(stderr, "[gas window %s] %d of %d workgroups handed back: temperature %d, pressure %d, regimes %d, chunk form %d, eta %d\n",
                 what, h[8], nblk, h[0], h[1], h[2], h[3], h[4]);
#if RRX_GW_TIMING
    unsigned long long clk[8], zero[8] = {0};
    if (hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_gw_clk), sizeof(clk)) == hipSuccess)
    {
        std::fprintf(stderr, "[gas window %s] clocks per workgroup (wavefront 0): set-up %.0f, chunk prologue %.0f, staging %.0f, dma wait %.0f, barriers %.0f, g-point loop %.0f, tail %.0f, contributor scalings %.0f\n",
                     what, double(clk[0])/nblk, double(clk[1])/nblk, double(clk[2])/nblk, double(clk[3])/nblk, double(clk[4])/nblk, double(clk[5])/nblk, double(clk[6])/nblk, double(clk[7])/nblk);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gw_clk), zero, sizeof(zero));
    }
#endif
}

template<typename F> struct PlanckArgs
{
    const F* pfracin; const F* tlev; const F* tsfc; int sfc_lay; int nPlanckTemp; const int* gpoint_bands;
    F totplnk_delta; const F* totplnk; F* pfrac; F* blay; F* blev; F* sfc_src; F* sfc_src_jac;
};

// Index tables of the windowed kernel (ints; a multiple of four: they travel as 16-byte words). The g-point loop of a workgroup
// runs chunk by chunk; a chunk is a run of at most GCH g-points inside which nothing changes -- neither the flavor of either regime
// nor the set of minor contributors -- so chunks end where bands end (16-g-point bands: the chunks of rounds 1-3; the 8-g-point
// bands of the reduced k-distributions, g128 / g112: chunks of 8 -- cut every 16 g-points regardless of the bands, 24 % of their
// workgroups were handed back and the rest ran at half the rate). `ncmax` bounds the number of chunks (host: ngpt/16 + nband).
struct GasWindowTables
{
    int ngpt, nmax, ncmax;
    static constexpr int LIT = 1 + NXW;                      // per chunk and regime: count (uncapped), then up to NXW contributor indices
    __host__ __device__ int off_cinfo() const { return 2*ngpt; }                         // [0] chunks, [1] regular, [2 + c] first g-point of chunk c (ncmax + 1)
    __host__ __device__ int off_lists() const { return off_cinfo() + ncmax + 3; }        // [2][ncmax][LIT]
    __host__ __device__ int off_mmeta() const { return off_lists() + 2*ncmax*LIT; }      // [2][nmax][MM]
    __host__ __device__ int off_cuni()  const { return off_mmeta() + 2*MM*nmax; }        // [2][ncmax] chunk usable by the windowed path
    __host__ __device__ int off_order() const { return off_cuni() + 2*ncmax; }           // [2][ncmax] chunk order (flavor by flavor)
    __host__ __device__ int ints()      const { return (off_order() + 2*ncmax + 3) & ~3; }
};
inline int gas_window_ncmax(const int ngpt, const int nband) { return (ngpt + GCH - 1) / GCH + std::max(nband, 0); }

// The tables depend on the k-distribution alone. One small workgroup builds them; the 9 000 workgroups of the windowed
// kernel copy a few KB instead of each walking the contributor arrays (set-up 0.40 -> 0.30 of 3.5 ms at C4).
// Round 4: the buffer persists between launches (gas_window_tables below). The kernel reads the index arrays it depends on (as
// before), compares them and the shape with what the buffer was built from -- both are part of the tables -- and leaves when
// nothing changed: 39 -> 5 us per launch, which was 4 % of a step at 2 048 columns per GPU. A k-distribution that changed in
// place, or another one at the same addresses, is rebuilt: the check is on the contents, not on the pointers.
constexpr int GW_TBL_HEADER = 8;                 // ints behind the tables: magic, ngpt, nminorlower, nminorupper, ncmax, nlist
constexpr int GW_TBL_MAGIC = 0x52525834;
__global__ void __launch_bounds__(256)
gas_window_tables_kernel(
        const int ngpt, const int nminorlower, const int nminorupper, const int ncmax, const int nlist,
        const int* __restrict__ gpoint_flavor,
        const int* __restrict__ minor_limits_gpt_lower, const int* __restrict__ minor_limits_gpt_upper,
        const Bool* __restrict__ minor_scales_with_density_lower, const Bool* __restrict__ minor_scales_with_density_upper,
        const Bool* __restrict__ scale_by_complement_lower, const Bool* __restrict__ scale_by_complement_upper,
        const int* __restrict__ idx_minor_lower, const int* __restrict__ idx_minor_upper,
        const int* __restrict__ idx_minor_scaling_lower, const int* __restrict__ idx_minor_scaling_upper,
        const int* __restrict__ kminor_start_lower, const int* __restrict__ kminor_start_upper,
        int* __restrict__ tbl)
{
    extern __shared__ int lds_int[];
    const int nmax = max(nminorlower, nminorupper);
    const GasWindowTables T{ngpt, nmax, ncmax};
    constexpr int LIT = GasWindowTables::LIT;
    int* gflav = lds_int;                                   // [2][ngpt]
    int* cinfo = lds_int + T.off_cinfo();
    int* lists = lds_int + T.off_lists();
    int* mmeta = lds_int + T.off_mmeta();
    int* cuni = lds_int + T.off_cuni();
    int* order = lds_int + T.off_order();
    int* cut = lds_int + T.ints();                          // [ngpt + 1] scratch: 1 where a chunk must start
    const int tid = threadIdx.x;
    const int ntab = T.ints();
    for (int w = tid; w < ntab + ngpt + 4 + 2*((ngpt + 63)/64); w += 256) lds_int[w] = 0;
    __syncthreads();
    {
        for (int w = tid; w < 2*ngpt; w += 256) gflav[(w & 1)*ngpt + (w >> 1)] = gpoint_flavor[w] - 1;
        for (int w = tid; w < nminorlower; w += 256)
        {
            int* m = mmeta + MM*w;
            m[0] = idx_minor_lower[w]; m[1] = minor_scales_with_density_lower[w] ? 1 : 0;
            m[2] = idx_minor_scaling_lower[w]; m[3] = scale_by_complement_lower[w] ? 1 : 0;
            m[4] = minor_limits_gpt_lower[2*w]; m[5] = minor_limits_gpt_lower[2*w+1]; m[6] = kminor_start_lower[w];
        }
        for (int w = tid; w < nminorupper; w += 256)
        {
            int* m = mmeta + MM*(nmax + w);
            m[0] = idx_minor_upper[w]; m[1] = minor_scales_with_density_upper[w] ? 1 : 0;
            m[2] = idx_minor_scaling_upper[w]; m[3] = scale_by_complement_upper[w] ? 1 : 0;
            m[4] = minor_limits_gpt_upper[2*w]; m[5] = minor_limits_gpt_upper[2*w+1]; m[6] = kminor_start_upper[w];
        }
    }
    __syncthreads();
    // ---- built from the same inputs before? (flavors and contributor metadata sit in the tables as they were read)
    {
        int* head = tbl + ntab;
        int differs = (head[0] != GW_TBL_MAGIC || head[1] != ngpt || head[2] != nminorlower || head[3] != nminorupper
                       || head[4] != ncmax || head[5] != nlist) ? 1 : 0;
        for (int w = tid; w < 2*ngpt && !differs; w += 256) differs = (tbl[w] != gflav[w]) ? 1 : 0;
        for (int w = tid; w < 2*MM*nmax && !differs; w += 256) differs = (tbl[T.off_mmeta() + w] != mmeta[w]) ? 1 : 0;
        if (!__syncthreads_or(differs)) return;
        if (tid == 0) head[0] = 0;                         // (not valid while it is being rewritten)
    }
    // ---- where chunks must start: a flavor change in either regime, the first g-point of a contributor's interval, the g-point
    // behind its last
    for (int g = tid; g < ngpt; g += 256)
        if (g > 0 && (gflav[g] != gflav[g-1] || gflav[ngpt + g] != gflav[ngpt + g-1])) cut[g] = 1;
    for (int w = tid; w < 2*nmax; w += 256)
    {
        const int r = w / nmax, i = w % nmax;
        if (i < (r == 0 ? nminorlower : nminorupper))
        {
            const int* m = mmeta + MM*w;
            const int lo = m[4]-1, hi = m[5];
            if (lo > 0 && lo < ngpt) cut[lo] = 1;
            if (hi > 0 && hi < ngpt) cut[hi] = 1;
        }
    }
    __syncthreads();
    // (the cut flags of 64 g-points as one ballot word each, so that one thread can walk over the cuts instead of over the g-points)
    unsigned long long* cutmask = reinterpret_cast<unsigned long long*>(cut + ((ngpt + 2) & ~1));      // [(ngpt + 63)/64]
    for (int base = 0; base < ngpt; base += 256)
    {
        const int g = base + tid;
        const unsigned long long m = __ballot(g > 0 && g < ngpt && cut[g] != 0);
        if ((tid & 63) == 0 && base + (tid & ~63) < ngpt) cutmask[(base + tid) >> 6] = m;
    }
    __syncthreads();
    if (tid == 0)
    {
        int n = 0, start = 0;
        bool fits = true;
        auto emit_until = [&](const int p)                   // chunks of at most GCH g-points from `start` up to the cut at p
        {
            while (start < p)
            {
                if (n < ncmax) cinfo[2 + n] = start; else fits = false;
                ++n; start = min(start + GCH, p);
            }
        };
        for (int w = 0; w < (ngpt + 63)/64; ++w)
        {
            unsigned long long m = cutmask[w];
            while (m != 0ull) { const int b = __ffsll((long long)m) - 1; m &= m - 1ull; emit_until(64*w + b); }
        }
        emit_until(ngpt);
        if (!fits || !RRX_GW_BANDCHUNKS)                     // (more runs than the bound allows for: the plain 16-g-point cut; what does not
        {                                                    //  fit the staged form there is handed back, as in rounds 1-3)
            n = (ngpt + GCH - 1) / GCH;
            for (int c=0; c<n; ++c) cinfo[2 + c] = c*GCH;
        }
        cinfo[2 + n] = ngpt;
        cinfo[0] = n;
        bool regular = true;
        for (int c=0; c<n; ++c) regular = regular && (cinfo[2 + c] == c*GCH);
        cinfo[1] = regular ? 1 : 0;
    }
    __syncthreads();
    const int nchunk = cinfo[0];
    // per-chunk contributor lists (ascending index = the reference's summation order) and the usability flag of the chunk:
    // one flavor over the chunk, every contributor on that flavor, at most `nlist` of them (NXW; NCW where the form of the launch has
    // no pass for the later ones)
    for (int w = tid; w < 2*nchunk; w += 256)
    {
        const int r = w / nchunk, c = w % nchunk;
        const int n = r == 0 ? nminorlower : nminorupper;
        const int c0 = cinfo[2 + c], c1 = cinfo[3 + c];
        int* out = lists + (r*ncmax + c)*LIT;
        const int fl = gflav[r*ngpt + c0];
        bool ok = true;
        for (int ig=c0+1; ig<c1; ++ig) ok = ok && (gflav[r*ngpt + ig] == fl);
        int cnt = 0;
        for (int i=0; i<n; ++i)
        {
            const int* m = mmeta + MM*(r*nmax + i);
            const int lo = m[4]-1, hi = m[5];
            if (lo < c1 && hi > c0)
            {
                if (cnt < nlist) { out[1 + cnt] = i; ok = ok && (gflav[r*ngpt + lo] == fl); }
                ++cnt;
            }
        }
        out[0] = cnt;
        cuni[r*ncmax + c] = (ok && cnt <= nlist) ? 1 : 0;
    }
    // chunk order per regime: chunks of one flavor next to each other (stable), so that a workgroup evaluates each flavor's
    // interpolation state once instead of once per band that uses it
    if (tid < 2)
    {
        int* ord = order + tid*ncmax;
        int k = 0;
        for (int c=0; c<nchunk; ++c)
        {
            const int fl = gflav[tid*ngpt + cinfo[2 + c]];
            bool seen = false;
            for (int d=0; d<c; ++d) seen = seen || (gflav[tid*ngpt + cinfo[2 + d]] == fl);
            if (seen) continue;
            for (int d=c; d<nchunk; ++d) if (gflav[tid*ngpt + cinfo[2 + d]] == fl) ord[k++] = d;
        }
    }

    __syncthreads();
#ifdef RRX_GW_DEBUG_TABLES
    if (tid == 0)
    {
        printf("tables: ngpt %d ncmax %d chunks %d regular %d\n", ngpt, ncmax, cinfo[0], cinfo[1]);
        for (int c=0; c<nchunk; ++c)
            printf("  chunk %d [%d,%d) lower: n %d usable %d fl %d | upper: n %d usable %d fl %d\n", c, cinfo[2+c], cinfo[3+c],
                   lists[c*LIT], cuni[c], gflav[cinfo[2+c]], lists[(ncmax + c)*LIT], cuni[ncmax + c], gflav[ngpt + cinfo[2+c]]);
    }
#endif
    for (int w = tid; w < ntab; w += 256) tbl[w] = lds_int[w];
    __threadfence();
    __syncthreads();
    if (tid == 0)
    {
        int* head = tbl + ntab;
        head[1] = ngpt; head[2] = nminorlower; head[3] = nminorupper; head[4] = ncmax; head[5] = nlist;
        __threadfence();
        head[0] = GW_TBL_MAGIC;
    }
}

// The persistent table buffer of a k-distribution: one per (calling thread, device, first index array, shape), zeroed when it is
// made, validated against the index arrays' CONTENTS by the kernel at every launch. At most 32 are kept (oldest dropped).
inline int* gas_window_tables(hipStream_t st, const int* key_ptr, const int ngpt, const int nminorlower, const int nminorupper,
                              const int ncmax, const int nlist)
{
    struct Entry { int dev; const int* key; int dims[5]; int* buf; };
    static thread_local std::vector<Entry> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) throw std::runtime_error("no device");
    const int dims[5] = {ngpt, nminorlower, nminorupper, ncmax, nlist};
    for (const Entry& e : cache)
        if (e.dev == dev && e.key == key_ptr && std::equal(dims, dims + 5, e.dims)) return e.buf;
    (void)st;
    // (plain hipMalloc / hipMemset, once per k-distribution: the buffer may be used from any stream of this thread afterwards)
    if (cache.size() >= 32) { (void)hipFree(cache.front().buf); cache.erase(cache.begin()); }
    const size_t n = size_t(GasWindowTables{ngpt, std::max(nminorlower, nminorupper), ncmax}.ints()) + GW_TBL_HEADER;
    Entry e{dev, key_ptr, {dims[0], dims[1], dims[2], dims[3], dims[4]}, nullptr};
    if (hipMalloc(reinterpret_cast<void**>(&e.buf), n*sizeof(int)) != hipSuccess) throw std::runtime_error("table allocation failed");
    if (hipMemset(e.buf, 0, n*sizeof(int)) != hipSuccess) throw std::runtime_error("memset failed");
    cache.push_back(e);
    return e.buf;
}

template<typename F>
size_t gas_window_lds_bytes(const int ngpt, const int nmax, const int ncmax, const int mode, const bool pf)
{
    const size_t ints = size_t(GasWindowTables{ngpt, nmax, ncmax}.ints()) + 16 + 8*size_t(ncmax);      // tables, reductions, per-chunk bands and key species
    const size_t pairs = size_t(GCH)*WBOX*(pf ? 2 : 1) + size_t(NCW)*GCH*MBOX + (mode == 1 ? size_t(GCH)*MBOX : 0);
    return ((ints*sizeof(int) + 15) & ~size_t(15)) + pairs*2*sizeof(F);
}

#ifndef RRX_GW_MINW
#define RRX_GW_MINW 3
#endif
template<typename F, int MODE, bool PF, bool CLD = false>
__global__ void __launch_bounds__(256, RRX_GW_MINW)
gas_window_kernel(
        const int ncol, const int nlay, const int ngpt, const int neta, const int npres, const int ntemp,
        const int nminorlower, const int nminorupper, const int idx_h2o,
        const int* __restrict__ gpoint_flavor,
        const F* __restrict__ kmajor, const F* __restrict__ kminor_lower, const F* __restrict__ kminor_upper,
        const int* __restrict__ minor_limits_gpt_lower, const int* __restrict__ minor_limits_gpt_upper,
        const Bool* __restrict__ minor_scales_with_density_lower, const Bool* __restrict__ minor_scales_with_density_upper,
        const Bool* __restrict__ scale_by_complement_lower, const Bool* __restrict__ scale_by_complement_upper,
        const int* __restrict__ idx_minor_lower, const int* __restrict__ idx_minor_upper,
        const int* __restrict__ idx_minor_scaling_lower, const int* __restrict__ idx_minor_scaling_upper,
        const int* __restrict__ kminor_start_lower, const int* __restrict__ kminor_start_upper,
        const F* __restrict__ play, const F* __restrict__ tlay, const F* __restrict__ col_gas, const F* __restrict__ col_dry,
        const F* __restrict__ krayl, const InterpArgs<F> ia,
        F* __restrict__ tau, F* __restrict__ ssa, F* __restrict__ g, const PlanckArgs<F> pa,
        int* __restrict__ todo, const int geom, const int* __restrict__ tbl, const int ncmax)
{
    // The product chain's kernel: multiply-adds of the node sums are contracted into FMAs here and the single-scattering albedo
    // uses a Newton reciprocal (one rounding fewer per term: 1e-15 relative from the gather / reference-shaped kernels, which
    // stay bit-exact against the goldens; tests hold this kernel to 1e-12). The interpolation state (cell_state, flavor_state:
    // integer indices depend on it) and the by-band cloud combination are separate functions and keep their rounding.
    #pragma clang fp contract(fast)
    typedef F Vec2 __attribute__((ext_vector_type(2)));
    typedef F Vec2u __attribute__((ext_vector_type(2), aligned(sizeof(F))));
    (void)sizeof(Vec2u);
    extern __shared__ int lds_int[];
    const int nmax = max(nminorlower, nminorupper);
    const GasWindowTables T{ngpt, nmax, ncmax};             // layout of the copied tables (gas_window_tables_kernel)
    constexpr int LIT = GasWindowTables::LIT;
    int* gflav = lds_int;                                   // [2][ngpt]
    const int* cinfo = lds_int + T.off_cinfo();             // [0] chunks, [1] regular (every chunk starts at a multiple of GCH), [2 + c] first g-point of chunk c
    const int* cstart = cinfo + 2;
    int* lists = lds_int + T.off_lists();                   // [2][ncmax][LIT]: count (uncapped), then up to NCW contributor indices
    int* mmeta = lds_int + T.off_mmeta();                   // [2][nmax][MM]
    int* cuni = lds_int + T.off_cuni();                     // [2][ncmax]: chunk usable by the windowed path (per regime)
    int* red = lds_int + T.ints();                          // [16] workgroup reductions (behind the copied tables)
    // per-chunk band numbers of the fractions form and of the by-band (all-sky) properties, looked up ONCE at set-up: read from global
    // memory at the top of a chunk (round 3) each look-up was a dependent load that queues behind the stores of the chunk before --
    // 3-5 k clocks, and the all-sky band search made several of them per chunk (phase clocks: chunk prologue +45 ... +85 k per workgroup)
    int* cband = red + 16;                                  // [ncmax][8]: PF first band, PF last band, CLD band, CLD chunk in one band,
                                                            //             key species (gas1, gas2) of the chunk's flavor in the lower / upper atmosphere
    const size_t int_bytes = ((size_t(T.ints()) + 16 + 8*size_t(ncmax))*sizeof(int) + 15) & ~size_t(15);
    Vec2* Wmaj = reinterpret_cast<Vec2*>(reinterpret_cast<char*>(lds_int) + int_bytes);     // [GCH][WBOX]
    Vec2* Wpf  = Wmaj + GCH*WBOX;                                                           // [GCH][WBOX] (PF)
    Vec2* Wmin = Wpf + (PF ? GCH*WBOX : 0);                                                 // [NCW][GCH][MBOX]
    Vec2* Wray = Wmin + NCW*GCH*MBOX;                                                       // [GCH][MBOX] (MODE 1)

    const int tid = threadIdx.y*64 + threadIdx.x;
#if RRX_GW_TIMING
    unsigned long long gw_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gw_t = __builtin_readcyclecounter();
#endif
    // the index tables depend on the k-distribution alone: gas_window_tables_kernel built them once for this launch
    {
        const int n4 = T.ints() / 4;
        const int4* __restrict__ src = reinterpret_cast<const int4*>(tbl);
        int4* dst = reinterpret_cast<int4*>(lds_int);
        for (int w = tid; w < n4; w += 256) dst[w] = src[w];
        if (tid < 16) red[tid] = (tid >= 6) ? 0 : ((tid & 1) ? -2147483647 : 2147483647);   // 0..5: running min / max pairs; 6, 8: presence masks
    }

    // geom 0: the workgroup's four waves are four consecutive layers of 64 columns; geom 1: four 64-column stretches of ONE layer
    // (gas_window_geometry: no workgroup straddles the tropopause or an eta jump between layers then)
    const int icol_raw = geom ? (blockIdx.x*4 + threadIdx.y)*64 + threadIdx.x : blockIdx.x*64 + threadIdx.x;
    const int ilay_raw = geom ? int(blockIdx.y) : blockIdx.y*4 + threadIdx.y;
    const bool active = icol_raw < ncol && ilay_raw < nlay;
    const int icol = min(icol_raw, ncol-1), ilay = min(ilay_raw, nlay-1);     // inactive threads shadow a valid cell (no stores)
    const size_t ncl = size_t(ncol)*nlay;
    const size_t idx = icol + size_t(ilay)*ncol;
    const F pl = play[idx], tl = tlay[idx];
    const CellState<F> cs = cell_state<F>(ia, npres, ntemp, pl, tl);
    const int itr = cs.itropo;
    const int jt = cs.jt, jp = cs.jp_raw + itr;

    // ---- box in temperature and pressure, one regime per workgroup
    __syncthreads();                                  // the tables and the reduction slots are in place
    atomicMin(&red[0], jt); atomicMax(&red[1], jt); atomicMin(&red[2], jp); atomicMax(&red[3], jp);
    atomicMin(&red[4], itr); atomicMax(&red[5], itr);
    if (tid < cinfo[0])
    {
        const int c0_ = cstart[tid], ge_ = cstart[tid+1];
        if constexpr (PF) { cband[8*tid] = pa.gpoint_bands[c0_] - 1; cband[8*tid+1] = pa.gpoint_bands[ge_-1] - 1; }
        if constexpr (CLD)
        {
            int b0 = 0;
            while (c0_ + 1 > ia.cld_lims[2*b0+1]) ++b0;
            cband[8*tid+2] = b0; cband[8*tid+3] = (ge_ <= ia.cld_lims[2*b0+1]) ? 1 : 0;
        }
        #pragma unroll
        for (int r=0; r<2; ++r)
        {
            const int fl_ = gflav[r*ngpt + c0_];
            cband[8*tid+4+2*r] = ia.flavor[2*fl_]; cband[8*tid+5+2*r] = ia.flavor[2*fl_+1];
        }
    }
    __syncthreads();
    const int jt_lo = red[0], jp_lo = red[2];
    // grid.z parts share out the chunks of a workgroup when the (column, layer) grid alone leaves CUs idle (few columns per GPU).
    // A part is a range of 16-g-point stretches, which the gather kernel can redo from the part's number alone; where the chunks do
    // not start at multiples of 16 (band-aligned chunks of a reduced k-distribution) part 0 takes them all and hands back "the
    // whole range" (part = gridDim.z).
    const int nchunk = rfl(cinfo[0]);
    const bool whole_range = gridDim.z > 1 && rfl(cinfo[1]) == 0;
    if (whole_range && blockIdx.z > 0) return;
    const int c_per = whole_range ? nchunk : (nchunk + int(gridDim.z) - 1) / int(gridDim.z);
    const int c_lo = whole_range ? 0 : int(blockIdx.z)*c_per, c_hi = min(nchunk, c_lo + c_per);
    bool fits = (red[1] - jt_lo < NTW) && (red[3] - jp_lo + 2 <= NPW) && (red[4] == red[5]);
    {
        bool all_chunks = true;
        for (int c=c_lo; c<c_hi; ++c) all_chunks = all_chunks && (cuni[itr*ncmax + c] != 0);
        fits = fits && all_chunks;
    }
    // workgroup-uniform: the gather kernel redoes this workgroup from scratch. The eight words in front of the list count the
    // reasons (0 temperature spread, 1 pressure spread, 2 both regimes, 3 a chunk outside the staged form, 4 eta spread);
    // RRX_GW_STATS=1 prints them
    auto hand_back = [&](const int why)
    {
        if (tid == 0)
        {
            const int k = atomicAdd(&todo[0], 1);
            todo[1 + k] = int(blockIdx.y*gridDim.x + blockIdx.x) + int(whole_range ? gridDim.z : blockIdx.z)*int(gridDim.x*gridDim.y);
            atomicAdd(&todo[why - 8], 1);
        }
    };
    if (!fits)
    {
        hand_back(!(red[1] - jt_lo < NTW) ? 0 : !(red[3] - jp_lo + 2 <= NPW) ? 1 : (red[4] != red[5]) ? 2 : 3);
        return;
    }

    const int tn = ntemp*neta;
    const size_t s_gpt = size_t(ntemp)*neta*(npres+1);
    constexpr unsigned SZ = sizeof(F);
    const F cdry0 = col_gas[idx];
    const F ch2o = col_gas[idx + size_t(idx_h2o)*ncl];
    F ray_fac = F(0.);
    if constexpr (MODE == 1) ray_fac = ch2o + col_dry[idx];
    const int ti = jt - jt_lo;                                        // pair (jt-1, jt) inside the box
    const int pi0 = jp - jp_lo;                                       // pressure node jp-1 inside the box (box starts at jp_lo-1)

    // a cell's element of g-point slab ig of a (col, lay, gpt) array: uniform 64-bit slab base + the cell's 32-bit byte offset --
    // the scalar-base form of the store (no 64-bit address arithmetic per lane and store, no address registers)
    const unsigned idx_b = unsigned(idx)*SZ;
    auto slab_store = [&](F* __restrict__ arr, const int ig, const F v)
    {
        stream_store(reinterpret_cast<F*>(reinterpret_cast<char*>(arr + size_t(ig)*ncl) + idx_b), v);
    };
    auto minor_scaling = [&](const int imnr) -> F                      // gas_optics_rrtmgp_kernels.cu:505-529
    {
        const int* m = mmeta + MM*(itr*nmax + imnr);
        const int imn = rfl(m[0]), swd = rfl(m[1]), ims = rfl(m[2]), sbc = rfl(m[3]);
        F scaling = col_gas[idx + size_t(imn)*ncl];
        const F cscal = col_gas[idx + size_t(max(ims, 0))*ncl];
        if (swd)
        {
            scaling *= F(0.01) * pl / tl;
            if (ims > 0)
            {
                const F vmr_fact = F(1.) / cdry0;
                const F dry_fact = F(1.) / (F(1.) + ch2o * vmr_fact);
                const F x = cscal * vmr_fact * dry_fact;
                scaling *= sbc ? (F(1.) - x) : x;
            }
        }
        return scaling;
    };

    // Planck-lite extras
    [[maybe_unused]] F b_sfc = F(0.), b_sfc2 = F(0.);     // (the surface temperature is re-read at each band change, by the surface layer's lanes)
    [[maybe_unused]] bool is_last = false, is_sfc = false;
    [[maybe_unused]] int cur_bnd = -1;
    [[maybe_unused]] const size_t ncv = size_t(ncol)*(nlay+1);
    [[maybe_unused]] F t_lev = F(0.);                      // kept for the whole kernel: a band step then has one memory round trip, not two
    if constexpr (PF)
    {
        is_last = ilay == nlay-1; is_sfc = ilay == pa.sfc_lay-1;
        t_lev = pa.tlev[idx];
    }
    // Fractions form: the band-integrated Planck functions B_lay, B_lev of the bands of this workgroup's chunks, all at once, HERE --
    // before the workgroup has a store in flight. Evaluated band by band at the top of each chunk (rounds 2-3) every band cost a
    // memory round trip behind the stores of the chunk before (phase clocks: chunk prologue 104 k clocks per workgroup against 41 k
    // in the SW form) and two fp64 divisions; now the table positions are found once and all bands' reads are in flight together.
    if constexpr (PF)
    {
        const int b_first = rfl(cband[8*c_lo]), b_last = rfl(cband[8*(c_hi-1)+1]);
        const size_t ncv_ = size_t(ncol)*(nlay+1);
        const Interp1dPos<F> p_lay = interp1d_pos(tl, ia.temp_ref_min, pa.totplnk_delta, pa.nPlanckTemp);
        const Interp1dPos<F> p_lev = interp1d_pos(t_lev, ia.temp_ref_min, pa.totplnk_delta, pa.nPlanckTemp);
        Interp1dPos<F> p_levp = p_lev;
        if (is_last) p_levp = interp1d_pos(pa.tlev[idx + ncol], ia.temp_ref_min, pa.totplnk_delta, pa.nPlanckTemp);
        #pragma unroll 4
        for (int ib=b_first; ib<=b_last; ++ib)
        {
            const F* tp = pa.totplnk + size_t(ib)*pa.nPlanckTemp;
            const F bl = interp1d_at(p_lay, tp), bv = interp1d_at(p_lev, tp);
            if (active) { pa.blay[idx + size_t(ib)*ncl] = bl; pa.blev[idx + size_t(ib)*ncv_] = bv; }
            if (is_last && active) pa.blev[idx + ncol + size_t(ib)*ncv_] = interp1d_at(p_levp, tp);
        }
    }


    int cur_flav = -1, je_lo = 1;
    [[maybe_unused]] int ne_x = NEW;                                   // eta nodes of the current flavor's box that are in use
    [[maybe_unused]] int cb = 0, cb_have = -1;   // CLD: band (0-based) of the g-point being stored (g-points ascend over the chunk loop)
    [[maybe_unused]] F c_tau = F(0.), c_ssa = F(0.), c_g = F(0.);
    F fm[8], cm[2], fn[4]; int je[2] = {1, 1};
    #pragma unroll
    for (int i=0; i<8; ++i) fm[i] = F(0.);
    cm[0] = cm[1] = F(0.); fn[0] = fn[1] = fn[2] = fn[3] = F(0.);
    const F* kmin = itr == 0 ? kminor_lower : kminor_upper;
    int red_slot = 6;                                                  // alternating pairs of reduction slots: 6/7, 8/9

    if (RRX_GW_ABL == 1) return;
    RRX_GW_T(0)
    const int* corder = lds_int + T.off_order() + rfl(itr)*ncmax;      // (one regime per workgroup here)
    // per contributor of the chunk: first g-point, end, offset of its kminor rows (from the metadata table)
    auto item_meta = [&](const int* items, const int i, int& lo, int& hi, int& koff)
    {
        const int* m = mmeta + MM*(rfl(itr)*nmax + rfl(items[i]));
        lo = rfl(m[4]) - 1; hi = rfl(m[5]); koff = rfl(m[6]) - 1 - lo;
    };
    for (int kc=c_lo; kc<c_hi; ++kc)
    {
        const int c = (gridDim.z == 1 || whole_range) ? rfl(corder[kc]) : kc;   // (parts of the chunk range keep the natural order)
        const int c0 = rfl(cstart[c]), gend = rfl(cstart[c+1]), ng = gend - c0;
        const int fl = gflav[itr*ngpt + c0];
        if (fl != cur_flav)                                            // workgroup-uniform
        {
            cur_flav = fl;
            const int gas1 = rfl(cband[8*c+4+2*rfl(itr)]), gas2 = rfl(cband[8*c+5+2*rfl(itr)]);     // (= ia.flavor[2*fl], [2*fl+1], looked up at set-up)
            const F cg1 = col_gas[idx + size_t(gas1)*ncl], cg2 = col_gas[idx + size_t(gas2)*ncl];
            #pragma unroll
            for (int itemp=0; itemp<2; ++itemp)
            {
                F fmi[2], fma[4];
                flavor_state<F>(ia, cs, neta, itemp, gas1, gas2, cg1, cg2, cm[itemp], je[itemp], fmi, fma);
                fn[2*itemp] = fmi[0]; fn[2*itemp+1] = fmi[1];
                fm[4*itemp] = fma[0]; fm[4*itemp+1] = fma[1]; fm[4*itemp+2] = fma[2]; fm[4*itemp+3] = fma[3];
            }
            // eta box of this flavor over the workgroup: presence mask of the eta indices in use (bit j = some cell has je == j),
            // built per wavefront from ballots (scalar work) and merged with one LDS atomic per wavefront
            unsigned present = 0u;
            for (int j=1; j<neta; ++j)
                if (__ballot(je[0] == j || je[1] == j) != 0ull) present |= (1u << j);
            if (threadIdx.x == 0) atomicOr(reinterpret_cast<unsigned*>(&red[red_slot]), present);
            __syncthreads();
            const unsigned all_present = reinterpret_cast<unsigned*>(red)[red_slot];
            je_lo = __ffs(int(all_present)) - 1;
            const int je_hi = 31 - __clz(int(all_present));
            red_slot = (red_slot == 6) ? 8 : 6;
            if (tid == 0) red[red_slot] = 0;                            // visible after the next barrier
            if (je_hi - je_lo + 2 > NEW) { hand_back(4); return; }
            ne_x = je_hi - je_lo + 2;
        }
        // (the regime is the same in every lane here: readfirstlane moves the chunk's list into scalar registers, so that the
        //  contributor conditions of the g-point loop are scalar branches instead of exec-mask sequences)
        const int itr_s = rfl(itr);
        const int n_all = rfl(lists[(itr_s*ncmax + c)*LIT]);             // (at most NXW here: the chunk is usable)
        const int n = min(n_all, NCW);                                   // contributors of the g-point loop; the others follow behind it
        const int* items = lists + (itr_s*ncmax + c)*LIT + 1;            // contributor indices of the chunk

        // all-sky: the cell's by-band values, read once per chunk where the chunk lies in one band (the rule), requested HERE -- ahead
        // of the barrier and the staging, whose wait they share
        [[maybe_unused]] auto cld_arrived = [&]()
        {
            if constexpr (CLD)
            {
                if constexpr (MODE != 2) asm volatile("" : "+v"(c_tau), "+v"(c_ssa), "+v"(c_g));
                else asm volatile("" : "+v"(c_tau));
            }
        };
        [[maybe_unused]] auto cld_load = [&](const int ib)
        {
            if constexpr (CLD)
            {
                cb_have = ib;
                const size_t b = idx + size_t(ib)*ncl;
                c_tau = ia.cld_tau[b];
                if constexpr (MODE != 2) { c_ssa = ia.cld_ssa[b]; c_g = ia.cld_g[b]; }
            }
        };
        [[maybe_unused]] bool cld_one_band = false;
        if constexpr (CLD)
        {
            const int b0 = rfl(cband[8*c+2]);
            cb = b0;
            cld_one_band = rfl(cband[8*c+3]) != 0;
            if (cld_one_band && b0 != cb_have) cld_load(b0);
        }

        RRX_GW_T(1)
        __syncthreads();                        // the previous chunk's readers are done with the windows
        RRX_GW_T(4)
        // per-cell scalings of this chunk's contributors (registers). Their column amounts are requested NEXT TO the staging loads --
        // behind the DMA issue in that form, ahead of the staging loads in the others -- so that both share one memory round trip
        // (round 4; evaluated behind the staging they cost a round trip of their own behind the stores in flight: ~2 k clocks per chunk).
        F sc[NCW]; int slo[NCW], shi[NCW], skoff[NCW];
        auto chunk_scalings = [&]()
        {
            #pragma unroll
            for (int i=0; i<NCW; ++i)
            {
                sc[i] = F(0.); slo[i] = 0; shi[i] = 0; skoff[i] = 0;
                if (i < n) { sc[i] = minor_scaling(rfl(items[i])); item_meta(items, i, slo[i], shi[i], skoff[i]); }
            }
        };
        // ---- stage the boxes: pairs (T, T+1) are adjacent words of the tables (temperature is their fastest dimension)
        if (RRX_GW_ABL != 2)
        {
            // The loads of a box are issued together, before its LDS writes: one memory round trip per phase (major [+ Planck
            // fractions]; Rayleigh + contributors 0-2; contributors 3-5 where there are any) instead of one per loop iteration
            // and box. The loops have at most GCH*WBOX/256 = 3 and 1 iterations: unrolled, the pairs held in registers.
            static_assert((GCH*WBOX) % 256 == 0 && GCH*MBOX <= 256 && NCW == 6, "staging phases are written for these box sizes");
            constexpr int KMAJ = GCH*WBOX/256;
            const int nmaj = ng*WBOX, nmin = ng*MBOX;
            // (the DMA moves 16 B per lane: fp64 pairs; the fp32 build keeps the register path)
            constexpr bool DMA = RRX_GW_LDSDMA && sizeof(F) == 8;
            // (the fp64 forms without fractions keep the DMA: with the staging registers they spill, and a spill reload at the top of a
            //  chunk waits behind every store in flight)
            if constexpr (!DMA || PF) chunk_scalings();             // (register-staged forms: their loads go out ahead of the staging loads)
            if constexpr (RRX_GW_SPARSE && (PF || !DMA))
            {
                // Sparse staging (round 3): a full box is 4 pressure x 4 eta nodes x 3 temperature pairs per g-point, what the cells of
                // a workgroup reach is usually 2 x 2 x 1 (one layer of neighbouring columns). Every 16-byte pair pulls its 128-byte line
                // through the L1, so staging the full boxes moved ~100 KB per chunk and workgroup (phase clocks: ~3 000 clocks per chunk,
                // 10-12 % of a workgroup's life). Only the nodes in reach are loaded -- into the same places of the same boxes, so the
                // g-point loop does not change; they go through registers because the places are no longer consecutive.
                // (x / d for x < 2^16 and small d as a multiplication: exact with m = floor((2^32 - 1) / d) + 1)
                auto magic = [](const int d) -> unsigned { return 0xFFFFFFFFu / unsigned(d) + 1u; };
                auto divs = [](const int x, const int d, const unsigned m) -> int { return d == 1 ? x : int(__umulhi(unsigned(x), m)); };
                const int np_x = rfl(red[3]) - jp_lo + 2, nt_x = rfl(red[1]) - jt_lo + 1;          // pressure nodes, temperature pairs in use
                const int et = ne_x*nt_x, wb = np_x*et;
                const unsigned m_wb = magic(wb), m_et = magic(et), m_t = magic(nt_x);
                const int n_maj = ng*wb, n_min = ng*et;
                for (int q = tid; q < n_maj; q += 256)
                {
                    const int gi = divs(q, wb, m_wb), r = q - gi*wb;
                    const int p_ = divs(r, et, m_et), r2 = r - p_*et;
                    const int e = divs(r2, nt_x, m_t), t = r2 - e*nt_x;
                    const int it_ = min(jt_lo - 1 + t, ntemp-2), ie = min(max(je_lo - 1 + e, 0), neta-1), ip = min(max(jp_lo - 1 + p_, 0), npres);
                    const unsigned off = (unsigned(c0 + gi)*unsigned(s_gpt) + unsigned(it_ + ie*ntemp + ip*tn))*SZ;
                    const int slot = gi*WBOX + (p_*NEW + e)*NTW + t;
                    const Vec2 vm = *reinterpret_cast<const Vec2u*>(reinterpret_cast<const char*>(kmajor) + off);
                    [[maybe_unused]] Vec2 vp;
                    if constexpr (PF) vp = *reinterpret_cast<const Vec2u*>(reinterpret_cast<const char*>(pa.pfracin) + off);
                    Wmaj[slot] = vm;
                    if constexpr (PF) Wpf[slot] = vp;
                }
                if (tid < n_min)
                {
                    const int gi_m = divs(tid, et, m_et), r_m = tid - gi_m*et;
                    const int e = divs(r_m, nt_x, m_t), t = r_m - e*nt_x;
                    const int it_m = min(jt_lo - 1 + t, ntemp-2), ie_m = min(max(je_lo - 1 + e, 0), neta-1);
                    const int slot = gi_m*MBOX + e*NTW + t;
                    const unsigned roff = unsigned(it_m + ie_m*ntemp)*SZ;
                    const F* kmin_u = rfl(itr) == 0 ? kminor_lower : kminor_upper;
                    auto minor_node = [&](const int i) -> Vec2           // (interval and row offset: read once per chunk, by chunk_scalings above)
                    {
                        const int kg = min(max(c0 + gi_m, slo[i]), shi[i]-1);           // clamped: always a valid table row
                        return *reinterpret_cast<const Vec2u*>(reinterpret_cast<const char*>(kmin_u) + unsigned((kg + skoff[i])*tn)*SZ + roff);
                    };
                    Vec2 v[3]; [[maybe_unused]] Vec2 vray;
                    if constexpr (MODE == 1)
                        vray = *reinterpret_cast<const Vec2u*>(reinterpret_cast<const char*>(krayl + size_t(rfl(itr))*tn*ngpt) + unsigned((c0 + gi_m)*tn)*SZ + roff);
                    #pragma unroll
                    for (int i=0; i<3; ++i) if (i < n) v[i] = minor_node(i);
                    if constexpr (MODE == 1) Wray[slot] = vray;
                    #pragma unroll
                    for (int i=0; i<3; ++i) if (i < n) Wmin[i*GCH*MBOX + slot] = v[i];
                    if (n > 3)
                    {
                        #pragma unroll
                        for (int i=3; i<NCW; ++i) if (i < n) v[i-3] = minor_node(i);
                        #pragma unroll
                        for (int i=3; i<NCW; ++i) if (i < n) Wmin[i*GCH*MBOX + slot] = v[i-3];
                    }
                }
                RRX_GW_T(2)
                RRX_GW_T(3)
            }
            else if constexpr (DMA)
            {
            // LDS-DMA staging (round 3): every pair-node goes from the table straight into its LDS slot (`global_load_lds_dwordx4`:
            // per-lane source address, destination = a wave-uniform base + 16 B x lane -- the boxes are laid out linearly in the
            // thread index for exactly that). No staging registers, so all loads of a chunk's boxes are in flight together in
            // every form (the fractions form used to take its two major-type boxes one pair of loads at a time), and no LDS store
            // instructions. Slots beyond a partial last chunk are filled from the chunk's last g-point (always whole wavefronts: the
            // instruction takes its LDS base from the first active lane).
            const F* kmin_u = rfl(itr) == 0 ? kminor_lower : kminor_upper;      // (uniform: one regime per workgroup)
            auto byte_off = [](const F* __restrict__ base, const unsigned boff) { return reinterpret_cast<const F*>(reinterpret_cast<const char*>(base) + boff); };
            auto glds = [](const F* __restrict__ src, Vec2* dst)
            {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            };
            (void)nmaj; (void)nmin; (void)KMAJ;
            #pragma unroll
            for (int k=0; k<KMAJ; ++k)
            {
                const int q = tid + 256*k;
                const int gi = min(q / WBOX, ng-1), r = q % WBOX;
                const int p = r / (NEW*NTW), e = (r / NTW) % NEW, t = r % NTW;
                const int it_ = min(jt_lo - 1 + t, ntemp-2), ie = min(max(je_lo - 1 + e, 0), neta-1), ip = min(max(jp_lo - 1 + p, 0), npres);
                const unsigned off = (unsigned(c0 + gi)*unsigned(s_gpt) + unsigned(it_ + ie*ntemp + ip*tn))*SZ;     // bytes (the tables are far below 4 GB)
                glds(byte_off(kmajor, off), Wmaj + q);
                if constexpr (PF) glds(byte_off(pa.pfracin, off), Wpf + q);
            }
            if (tid < GCH*MBOX)                                             // wavefronts 0-2 in full
            {
                const int gi_m = min(tid / MBOX, ng-1), r_m = tid % MBOX;   // this thread's node of a minor / Rayleigh box
                const int it_m = min(jt_lo - 1 + r_m % NTW, ntemp-2), ie_m = min(max(je_lo - 1 + r_m / NTW, 0), neta-1);
                if constexpr (MODE == 1)
                    glds(byte_off(krayl + size_t(rfl(itr))*tn*ngpt, unsigned((c0 + gi_m)*tn + it_m + ie_m*ntemp)*SZ), Wray + tid);
                #pragma unroll
                for (int i=0; i<NCW; ++i)
                    if (i < n)
                    {
                        int lo, hi, koff; item_meta(items, i, lo, hi, koff);
                        const int kg = min(max(c0 + gi_m, lo), hi-1);       // clamped: always a valid table row
                        glds(byte_off(kmin_u, unsigned((kg + koff)*tn + it_m + ie_m*ntemp)*SZ), Wmin + i*GCH*MBOX + tid);
                    }
            }
            chunk_scalings();
            RRX_GW_T(2)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the DMA writes have landed in LDS (the barrier below publishes them)
            RRX_GW_T(3)
            }
            else
            {
            auto stage_major = [&](const F* __restrict__ table, Vec2* __restrict__ W)
            {
                Vec2 v[KMAJ];
                #pragma unroll
                for (int k=0; k<KMAJ; ++k)
                {
                    const int q = tid + 256*k;
                    if (q < nmaj)
                    {
                        const int gi = q / WBOX, r = q % WBOX;
                        const int p = r / (NEW*NTW), e = (r / NTW) % NEW, t = r % NTW;
                        const int it_ = min(jt_lo - 1 + t, ntemp-2), ie = min(max(je_lo - 1 + e, 0), neta-1), ip = min(max(jp_lo - 1 + p, 0), npres);
                        v[k] = *reinterpret_cast<const Vec2u*>(table + size_t(c0 + gi)*s_gpt + size_t(it_) + size_t(ie)*ntemp + size_t(ip)*tn);
                    }
                }
                #pragma unroll
                for (int k=0; k<KMAJ; ++k)
                {
                    const int q = tid + 256*k;
                    if (q < nmaj) W[q] = v[k];                                                  // gi*WBOX + r == q
                }
            };
            if constexpr (!PF) stage_major(kmajor, Wmaj);
            else
            {
                // (the fractions form runs at the register limit of three waves per SIMD -- batching its two major-type boxes
                //  costs 84 B of scratch per lane and time -- so they keep the rolled loop, one pair of loads in flight per iteration)
                for (int q = tid; q < nmaj; q += 256)
                {
                    const int gi = q / WBOX, r = q % WBOX;
                    const int p = r / (NEW*NTW), e = (r / NTW) % NEW, t = r % NTW;
                    const int it_ = min(jt_lo - 1 + t, ntemp-2), ie = min(max(je_lo - 1 + e, 0), neta-1), ip = min(max(jp_lo - 1 + p, 0), npres);
                    const size_t off = size_t(c0 + gi)*s_gpt + size_t(it_) + size_t(ie)*ntemp + size_t(ip)*tn;
                    Wmaj[q] = *reinterpret_cast<const Vec2u*>(kmajor + off);
                    Wpf[q] = *reinterpret_cast<const Vec2u*>(pa.pfracin + off);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int gi_m = tid / MBOX, r_m = tid % MBOX;                  // this thread's node of a minor / Rayleigh box
            const int it_m = min(jt_lo - 1 + r_m % NTW, ntemp-2), ie_m = min(max(je_lo - 1 + r_m / NTW, 0), neta-1);
            auto minor_node = [&](const int i) -> Vec2
            {
                int lo, hi, koff; item_meta(items, i, lo, hi, koff);
                const int kg = min(max(c0 + gi_m, lo), hi-1);                   // clamped: always a valid table row
                return *reinterpret_cast<const Vec2u*>(kmin + size_t(kg + koff)*tn + it_m + ie_m*ntemp);
            };
            if (tid < nmin)
            {
                Vec2 v[3]; [[maybe_unused]] Vec2 vray;
                if constexpr (MODE == 1)
                    vray = *reinterpret_cast<const Vec2u*>(krayl + size_t(itr)*tn*ngpt + size_t(c0 + gi_m)*tn + it_m + ie_m*ntemp);
                #pragma unroll
                for (int i=0; i<3; ++i) if (i < n) v[i] = minor_node(i);
                if constexpr (MODE == 1) Wray[tid] = vray;                                       // gi*MBOX + r == tid
                #pragma unroll
                for (int i=0; i<3; ++i) if (i < n) Wmin[(i*GCH + gi_m)*MBOX + r_m] = v[i];
                if (n > 3)
                {
                    #pragma unroll
                    for (int i=3; i<NCW; ++i) if (i < n) v[i-3] = minor_node(i);
                    #pragma unroll
                    for (int i=3; i<NCW; ++i) if (i < n) Wmin[(i*GCH + gi_m)*MBOX + r_m] = v[i-3];
                }
            }
            }
        }
        if (RRX_GW_ABL == 2) chunk_scalings();
        RRX_GW_T(7)
        __syncthreads();
        RRX_GW_T(4)

        // ---- the cell's corners inside the boxes
        const int e0 = je[0] - je_lo, e1 = je[1] - je_lo;               // eta node je-1 sits at box index je - je_lo
        const int m00 = (pi0*NEW + e0)*NTW + ti, m10 = (pi0*NEW + e1)*NTW + ti;      // kmajor: pressure node jp-1, eta node je-1
        const int q0 = e0*NTW + ti, q1 = e1*NTW + ti;                               // kminor / krayl
        const bool wave_same_eta = !__any(je[0] != je[1]);
        const bool wave_all_active = !__any(!active);

        // band Planck functions and surface terms of the fractions form, once per band. The band is looked up once per chunk where the
        // chunk lies in one band (the rule): a per-g-point look-up is a global load the wavefront waits for with `vmcnt(0)`, i.e.
        // behind every store it has in flight -- round-3 phase clocks: 1 660 clocks per g-point and wavefront with it, against
        // ~800 for the same loop without.
        [[maybe_unused]] auto band_update = [&](const int ibnd)
        {
            if constexpr (PF)
            {
                cur_bnd = ibnd;
                const F* tp = pa.totplnk + size_t(ibnd)*pa.nPlanckTemp;
                // (B_lay, B_lev of every band of this workgroup's chunks were written at set-up; the surface layer's workgroups -- one
                //  layer in 140 -- still look the surface terms up band by band)
                if (is_sfc)
                {
                    const F t_sfc = pa.tsfc[icol];
                    b_sfc  = interp1d(t_sfc        , ia.temp_ref_min, pa.totplnk_delta, pa.nPlanckTemp, tp);
                    b_sfc2 = interp1d(t_sfc + F(1.), ia.temp_ref_min, pa.totplnk_delta, pa.nPlanckTemp, tp);
                }
            }
        };
        if constexpr (CLD) cld_arrived();
        [[maybe_unused]] bool one_band = false;
        if constexpr (PF)
        {
            const int b0 = rfl(cband[8*c]), b1 = rfl(cband[8*c+1]);
            one_band = b0 == b1;
            if (one_band && b0 != cur_bnd) band_update(b0);
        }
        RRX_GW_T(1)

        // ---- the chunk's g-points, one per iteration (measured alternatives, all slower on MI355X: batches of 2-8 g-points
        // with their LDS reads issued together -- the registers cost the third wave per SIMD --, contributor reads preloaded
        // next to the major term's, per-contributor sweeps over the chunk, straight-line specialisations per contributor count)
        // One g-point: the reference's expression order. `U` g-points at once (RRX_GW_PAIR): the same expressions per g-point,
        // written side by side so that the LDS reads of both go out together and the two dependent fp64 chains interleave
        // (a single chain leaves the SIMD idle for most of each LDS round trip with three waves to cover it). The paired form
        // is taken when every contributor of the chunk spans the whole chunk (the rule: intervals are band-aligned), so it
        // needs no per-g-point range tests.
        // bit gi of cmask[i]: contributor i is present at g-point c0 + gi (its interval [slo, shi) cut to the chunk; 0 beyond the list)
        unsigned cmask[NCW];
        #pragma unroll
        for (int i=0; i<NCW; ++i)
        {
            const int lo = max(slo[i] - c0, 0), hi = min(shi[i] - c0, ng);
            cmask[i] = (i < n && hi > lo) ? (((hi >= 32) ? ~0u : ((1u << hi) - 1u)) & ~((1u << lo) - 1u)) : 0u;
        }
        // Scalar instructions are not free (tools/issue_mix_bench.hip: a wavefront's s_* instruction costs its SIMD 2.8-4 cycles of issue,
        // as much as a vector one; the loop executed ~45 of them per g-point next to ~50 vector ones). Hence: (i) whether the wavefront's
        // cells share one eta index is decided once per chunk, not at every read (SAME_: two copies of the loop); (ii) "is contributor i
        // present at this g-point" is one bit test on a per-chunk mask instead of two compares against the interval's limits.
        // (iii) the stores address their g-point slab through a uniform 64-bit pointer that walks along with the loop (two scalar adds
        // per array and g-point; the compiler's own form re-multiplied ig * ncol * nlay and added it to a 64-bit address per lane: nine
        // scalar and two 64-bit vector instructions); (iv) a wavefront whose lanes all own a cell (ACT_: every wavefront but those of
        // the last column block) runs a copy of the loop without the store predicate.
        const size_t slab_b = ncl*SZ;
        char* sb_tau = reinterpret_cast<char*>(tau) + size_t(c0)*slab_b;
        [[maybe_unused]] char* sb_ssa = (MODE != 2) ? reinterpret_cast<char*>(ssa) + size_t(c0)*slab_b : nullptr;
        [[maybe_unused]] char* sb_g = (MODE != 2 && CLD) ? reinterpret_cast<char*>(g) + size_t(c0)*slab_b : nullptr;
        [[maybe_unused]] char* sb_pf = nullptr;
        if constexpr (PF) sb_pf = reinterpret_cast<char*>(pa.pfrac) + size_t(c0)*slab_b;
        auto slab_put = [&](char* sbase, const int u, const F v)
        {
            stream_store_sbase<F>(sbase + size_t(u)*slab_b, idx_b, v);
        };
        // (v) FULL_: every contributor of the chunk spans the whole chunk (the rule: intervals are band-aligned) -- the list is walked
        // until it ends (one compare per contributor present, none for the places behind the last one); otherwise the masks.
        // (Straight-line copies per contributor count, no tests at all, cost registers whatever the read-ahead: fp32 91 -> 127-167
        //  VGPRs, fp64 47-180 spilled.)
        // FAST_ = (iv) and (v) together and the chunk inside one band (Planck functions, by-band cloud values): the rule. Everything
        // else -- a wavefront of the last column block, an interval that ends inside the chunk, a band boundary inside it -- takes the
        // one general copy of the loop.
        auto gstep = [&](auto U_, auto SAME_, auto FAST_, const int gi0)
        {
            constexpr int U = decltype(U_)::value;
            constexpr bool SAME = decltype(SAME_)::value;
            constexpr bool ACT = decltype(FAST_)::value, FULL = decltype(FAST_)::value, ONE = decltype(FAST_)::value;
            Vec2 a0[U], a1[U], a2[U], a3[U]; F k4[U], k5[U], k6[U], k7[U], t[U];
            #pragma unroll
            for (int u=0; u<U; ++u)
            {
                const Vec2* wm = Wmaj + (gi0 + u)*WBOX;
                // corners: lower temperature node (jt-1) = .x of the pair, upper (jt) = .y; the upper node uses its own eta index
                a0[u] = wm[m00]; a1[u] = wm[m00 + NTW]; a2[u] = wm[m00 + NEW*NTW]; a3[u] = wm[m00 + NEW*NTW + NTW];
                k4[u] = a0[u].y; k5[u] = a1[u].y; k6[u] = a2[u].y; k7[u] = a3[u].y;
            }
            if constexpr (!SAME)
            {
                #pragma unroll
                for (int u=0; u<U; ++u)
                {
                    const Vec2* wm = Wmaj + (gi0 + u)*WBOX;
                    k4[u] = wm[m10].y; k5[u] = wm[m10 + NTW].y; k6[u] = wm[m10 + NEW*NTW].y; k7[u] = wm[m10 + NEW*NTW + NTW].y;
                }
            }
            #pragma unroll
            for (int u=0; u<U; ++u)
                t[u] = cm[0] * (fm[0]*a0[u].x + fm[1]*a1[u].x + fm[2]*a2[u].x + fm[3]*a3[u].x)
                     + cm[1] * (fm[4]*k4[u] + fm[5]*k5[u] + fm[6]*k6[u] + fm[7]*k7[u]);
            #pragma unroll
            for (int i=0; i<NCW; ++i)                                   // ascending contributor index: the reference's order
            {
                if constexpr (FULL) { if (i >= n) break; }
                if (FULL || ((cmask[i] >> gi0) & 1u))
                {
                    Vec2 c0v[U], c1v[U]; F m2[U], m3[U];
                    #pragma unroll
                    for (int u=0; u<U; ++u)
                    {
                        const Vec2* wn = Wmin + (i*GCH + gi0 + u)*MBOX;
                        c0v[u] = wn[q0]; c1v[u] = wn[q0 + NTW]; m2[u] = c0v[u].y; m3[u] = c1v[u].y;
                    }
                    if constexpr (!SAME)
                    {
                        #pragma unroll
                        for (int u=0; u<U; ++u) { const Vec2* wn = Wmin + (i*GCH + gi0 + u)*MBOX; m2[u] = wn[q1].y; m3[u] = wn[q1 + NTW].y; }
                    }
                    #pragma unroll
                    for (int u=0; u<U; ++u)
                    {
                        const F kk = fn[0]*c0v[u].x + fn[1]*c1v[u].x + fn[2]*m2[u] + fn[3]*m3[u];
                        t[u] = t[u] + kk * sc[i];
                    }
                }
            }
            [[maybe_unused]] F ray[U];
            if constexpr (MODE != 2)
            {
                Vec2 r0[U], r1[U]; F r2[U], r3[U];
                #pragma unroll
                for (int u=0; u<U; ++u) { const Vec2* wr = Wray + (gi0 + u)*MBOX; r0[u] = wr[q0]; r1[u] = wr[q0 + NTW]; r2[u] = r0[u].y; r3[u] = r1[u].y; }
                if constexpr (!SAME)
                {
                    #pragma unroll
                    for (int u=0; u<U; ++u) { const Vec2* wr = Wray + (gi0 + u)*MBOX; r2[u] = wr[q1].y; r3[u] = wr[q1 + NTW].y; }
                }
                #pragma unroll
                for (int u=0; u<U; ++u) ray[u] = ray_fac * (fn[0]*r0[u].x + fn[1]*r1[u].x + fn[2]*r2[u] + fn[3]*r3[u]);
            }
            #pragma unroll
            for (int u=0; u<U; ++u)
            {
                const int gi = gi0 + u, ig = c0 + gi;
                if constexpr (CLD && !ONE)
                {
                    if (!cld_one_band)              // (a chunk with a band boundary inside: not the rule)
                    {
                        while (ig + 1 > rfl(ia.cld_lims[2*cb+1])) ++cb;
                        if (cb != cb_have) { cld_load(cb); cld_arrived(); }
                    }
                }
                if constexpr (MODE == 2)
                {
                    if constexpr (CLD) { if (ACT || active) slab_put(sb_tau, u, t[u] + c_tau); }
                    else if (ACT || active) slab_put(sb_tau, u, t[u]);
                }
                else
                {
                    F tt = t[u] + ray[u];
                    F ww = (tt > F(2.)*Lim<F>::eps()) ? ray[u] * fast_rcp(tt) : F(0.);
                    if constexpr (CLD)
                    {
                        F gg = F(0.);
                        add_cloud_to_gas_2str<F, RRX_GW_FAST_BYBAND != 0>(tt, ww, gg, c_tau, c_ssa, c_g);
                        if (ACT || active) { slab_put(sb_tau, u, tt); slab_put(sb_ssa, u, ww); slab_put(sb_g, u, gg); }
                    }
                    else if (ACT || active)
                    {
                        slab_put(sb_tau, u, tt);
                        slab_put(sb_ssa, u, ww);                // (a g array of a cloudless launch is zeroed behind the loop)
                    }
                }
                if constexpr (PF)
                {
                    const Vec2* wp = Wpf + gi*WBOX;
                    const Vec2 p0 = wp[m00], p1 = wp[m00 + NTW], p2 = wp[m00 + NEW*NTW], p3 = wp[m00 + NEW*NTW + NTW];
                    F v4 = p0.y, v5 = p1.y, v6 = p2.y, v7 = p3.y;
                    if constexpr (!SAME) { v4 = wp[m10].y; v5 = wp[m10 + NTW].y; v6 = wp[m10 + NEW*NTW].y; v7 = wp[m10 + NEW*NTW + NTW].y; }
                    const F pfrac = (fm[0]*p0.x + fm[1]*p1.x + fm[2]*p2.x + fm[3]*p3.x) + (fm[4]*v4 + fm[5]*v5 + fm[6]*v6 + fm[7]*v7);
                    if constexpr (!ONE)
                    {
                        if (!one_band)                              // (a chunk with a band boundary inside: not the rule)
                        {
                            const int ibnd = rfl(pa.gpoint_bands[ig]) - 1;
                            if (ibnd != cur_bnd) band_update(ibnd);
                        }
                    }
                    if (ACT || active)
                    {
                        slab_put(sb_pf, u, pfrac);
                        if (is_sfc)
                        {
                            pa.sfc_src    [icol + size_t(ig)*ncol] = pfrac * b_sfc;
                            pa.sfc_src_jac[icol + size_t(ig)*ncol] = pfrac * (b_sfc2 - b_sfc);
                        }
                    }
                }
            }
            sb_tau += U*slab_b;
            if constexpr (MODE != 2) sb_ssa += U*slab_b;
            if constexpr (MODE != 2 && CLD) sb_g += U*slab_b;
            if constexpr (PF) sb_pf += U*slab_b;
        };

        // every contributor of the chunk spans the chunk: no range tests, g-points can go in pairs
        bool chunk_full = true;
        #pragma unroll
        for (int i=0; i<NCW; ++i) if (i < n && !(slo[i] <= c0 && shi[i] >= gend)) chunk_full = false;
#ifndef RRX_GW_COUNTED
#define RRX_GW_COUNTED 1
#endif
#ifndef RRX_GW_PAIR32_SW
#define RRX_GW_PAIR32_SW 0
#endif
#ifndef RRX_GW_PAIR32
#define RRX_GW_PAIR32 1    // fp32 pairs the g-points of the fractions form only (LW stage 1.78 -> 1.69 ms at C4; SW 1.38 -> 1.49 paired)
#endif
#ifndef RRX_GW_PAIR_PF
#define RRX_GW_PAIR_PF 1
#endif
#ifndef RRX_GW_NOPAIR_CLD
#define RRX_GW_NOPAIR_CLD 1
#endif
        // (the fractions form and the all-sky SW form have no registers to spare in fp64: paired they spill, and a spill reload waits
        //  behind every store in flight)
        // (fp32: unpaired was faster in every form while the loop still carried its range tests -- LW stage 2.05 -> 1.87 ms, SW 1.56 -> 1.50 ms
        //  at C4; with the bit masks the fractions form gains from pairs, RRX_GW_PAIR32 above, the SW forms still do not)
        // (fractions form, fp64: paired it spills 72 B per lane and is still 3 % faster now that nothing in its loop waits on `vmcnt` --
        //  2.76 -> 2.68 ms, two boxes; before the band look-up left the loop it was 3 % slower. Not in the all-sky form.)
        constexpr int PAIR = (RRX_GW_PAIR && (sizeof(F) == 8 || (RRX_GW_PAIR32 && (PF || RRX_GW_PAIR32_SW))) && ((RRX_GW_PAIR_PF && !CLD) || !PF) && !(RRX_GW_NOPAIR_CLD && CLD && MODE == 1)) ? 2 : 1;
        auto gloop = [&](auto SAME_, auto FAST_)
        {
            int gi = 0;
            if constexpr (PAIR == 2)                                    // (pairs need no range tests: only where chunk_full)
            {
                if (decltype(FAST_)::value || chunk_full)
                    for (; gi + 1 < ng; gi += 2) gstep(std::integral_constant<int,2>{}, SAME_, FAST_, gi);
            }
            for (; gi < ng; ++gi) gstep(std::integral_constant<int,1>{}, SAME_, FAST_, gi);
        };
        if (RRX_GW_ABL != 3)
        {
            using T_ = std::true_type; using F_ = std::false_type;
            bool fast = RRX_GW_COUNTED && wave_all_active && chunk_full;
            if constexpr (CLD) fast = fast && cld_one_band;
            if constexpr (PF) fast = fast && one_band;
            if (fast)
            {
                if (wave_same_eta) gloop(T_{}, T_{}); else gloop(F_{}, T_{});
            }
            else
            {
                if (wave_same_eta) gloop(T_{}, F_{}); else gloop(F_{}, F_{});
            }
            if constexpr (MODE != 2 && !CLD)
            {
                if (g != nullptr && active)                             // asymmetry of a cloudless launch: zeros (stores only)
                    for (int gi=0; gi<ng; ++gi) slab_store(g, c0 + gi, F(0.));
            }
        }
        // ---- contributors beyond the NCW the boxes have room for (a band of the full gas set can have seven or more): one at a time,
        // its nodes staged into the first contributor box, its term added to the optical depths this chunk has just stored (and, in the
        // Rare and slow by design -- the alternative was to hand the whole regime back to the gather kernels. LW forms only: the
        // bands with seven contributors are longwave ones (1 080-1 180 cm-1 with the CFCs present), and in the SW form the extra code
        // cost the g-point loop three spilled registers (2.16 -> 2.22 ms, tools/ab_head.sh); its lists end at NCW.
        if constexpr (MODE == 2)
        if (n_all > NCW)                                                // workgroup-uniform
        {
            const F* kmin_u = rfl(itr) == 0 ? kminor_lower : kminor_upper;
            for (int x=NCW; x<n_all; ++x)
            {
                __syncthreads();                                        // the boxes of the loop (or of the contributor before) are free
                int xlo, xhi, xk; item_meta(items, x, xlo, xhi, xk);
                if (tid < GCH*MBOX)
                {
                    const int gi_m = min(tid / MBOX, ng-1), r_m = tid % MBOX;
                    const int it_m = min(jt_lo - 1 + r_m % NTW, ntemp-2), ie_m = min(max(je_lo - 1 + r_m / NTW, 0), neta-1);
                    const int kg = min(max(c0 + gi_m, xlo), xhi-1);     // clamped: always a valid table row
                    Wmin[tid] = *reinterpret_cast<const Vec2u*>(kmin_u + size_t(kg + xk)*tn + it_m + ie_m*ntemp);
                }
                const F scx = minor_scaling(rfl(items[x]));
                __syncthreads();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wavefront's stores of the chunk have left: they are read back
                for (int gi=0; gi<ng; ++gi)
                {
                    const int ig = c0 + gi;
                    if (ig < xlo || ig >= xhi) continue;                // (uniform)
                    const Vec2* wn = Wmin + gi*MBOX;
                    const Vec2 c0v = wn[q0], c1v = wn[q0 + NTW];
                    F m2 = c0v.y, m3 = c1v.y;
                    if (!wave_same_eta) { m2 = wn[q1].y; m3 = wn[q1 + NTW].y; }
                    const F add = (fn[0]*c0v.x + fn[1]*c1v.x + fn[2]*m2 + fn[3]*m3) * scx;
                    F* tp = reinterpret_cast<F*>(reinterpret_cast<char*>(tau + size_t(ig)*ncl) + idx_b);
                    if (active)
                    {
                        const F tt = __hip_atomic_load(tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;     // (past the L1)
                        slab_store(tau, ig, tt);
                    }
                }
            }
        }
        RRX_GW_T(5)
    }
#if RRX_GW_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RRX_GW_T(6)
    if (tid == 0) for (int k=0; k<8; ++k) atomicAdd(&g_gw_clk[k], gw_acc[k]);
#endif
    (void)SZ;
}

template<typename F>
__global__ void reorder123x321_kernel(const int ni, const int nj, const int nk, const F* __restrict__ in, F* __restrict__ out)
{
    const size_t n = size_t(ni)*nj*nk;
    for (size_t o = size_t(blockIdx.x)*blockDim.x + threadIdx.x; o < n; o += size_t(gridDim.x)*blockDim.x)
    {
        const int ii = int(o % ni), ij = int((o / ni) % nj), ik = int(o / (size_t(ni)*nj));
        out[o] = in[ik + size_t(ij)*nk + size_t(ii)*nj*nk];
    }
}

template<typename F>
__global__ void reorder12x21_kernel(const int ni, const int nj, const F* __restrict__ in, F* __restrict__ out)
{
    const size_t n = size_t(ni)*nj;
    for (size_t o = size_t(blockIdx.x)*blockDim.x + threadIdx.x; o < n; o += size_t(gridDim.x)*blockDim.x)
    {
        const int ii = int(o % ni), ij = int(o / ni);
        out[o] = in[ij + size_t(ii)*nj];
    }
}

// shared-cell path of the gather kernels (RRX_GO_SHARE=0 turns it off for A/B runs)

inline int grid1d(const size_t n) { return int(std::min<size_t>((n + 255)/256, 256*8)); }

template<typename F> size_t planck_lds_bytes(const int ngpt)
{
    return size_t(GCH)*(PL+1)*64*sizeof(F) + size_t((2*ngpt + 3) & ~3)*sizeof(int) + size_t(PL)*2*4*GCH*2*sizeof(F);
}

// LW gas optics + Planck-lite in one pass: windowed kernel with the Planck fractions riding along; the workgroups it hands
// back are finished by the gather kernel (tau) and planck_fraction_kernel (fractions, band Planck functions, surface terms)
template<typename F>
int gas_optics_lw_fractions_impl(
        int ncol, int nlay, int nband, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, int nPlanckTemp,
        int nminorlower, int nminorupper, int idx_h2o,
        const int* gpoint_flavor, const int* gpoint_bands,
        const F* kmajor, const F* kminor_lower, const F* kminor_upper,
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper,
        const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
        const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper,
        const int* idx_minor_lower, const int* idx_minor_upper,
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper,
        const int* kminor_start_lower, const int* kminor_start_upper,
        const InterpArgs<F> ia, const F* play, const F* tlay, const F* tlev, const F* tsfc, int sfc_lay, const F* col_gas,
        const F* pfracin, F totplnk_delta, const F* totplnk,
        F* tau, F* pfrac, F* blay, F* blev, F* sfc_src, F* sfc_src_jac, void* stream)
{
    RRX_TRY
    (void)ngas; (void)nflav;
    if (ncol <= 0 || nlay <= 0 || ngpt <= 0) throw std::runtime_error("empty problem");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nchunk = (ngpt + GCH - 1) / GCH;
    const int nmax = std::max(nminorlower, nminorupper);
    const size_t lds = (size_t(3)*ngpt + size_t(2)*nchunk*(1 + ITEM*nmax) + size_t(2)*MM*nmax)*sizeof(int);
    if (lds > 64*1024) throw std::runtime_error("minor-gas index exceeds 64 KiB of LDS");
    const dim3 block(64, 4);
    const dim3 grid(ceil_div(ncol, 64), ceil_div(nlay, 4));
    const int ncmax = gas_window_ncmax(ngpt, nband);
    const size_t wlds = gas_window_lds_bytes<F>(ngpt, nmax, ncmax, 2, true);
    // (the windowed kernel addresses a cell inside a g-point slab, and a node inside the kmajor / planck_frac tables, with 32-bit byte
    //  offsets; a table beyond 4 GB -- far above any k-distribution -- goes to the gather kernels instead of wrapping. The minor and
    //  Rayleigh tables are smaller than kmajor by the pressure dimension.)
    const bool windowed = tuning().go_window && wlds <= 64*1024 && size_t(ncol)*nlay*sizeof(F) < (size_t(1) << 32)
                          && size_t(ngpt)*ntemp*neta*(npres+1)*sizeof(F) < (size_t(1) << 32);
    StreamScratch scratch(st);
    const int geom = gas_window_geometry(ncol);
    const dim3 wgrid = gas_window_grid(geom, ncol, nlay);
    const int nblk = windowed ? int(wgrid.x)*int(wgrid.y) : int(grid.x)*int(grid.y);
    const int nz = gas_window_parts(nblk, nchunk);
    int* todo = nullptr;
    if (windowed)
    {
        todo = scratch.get<int>(size_t(9) + size_t(nblk)*nz) + 8;
        if (hipMemsetAsync(todo - 8, 0, 9*sizeof(int), st) != hipSuccess) throw std::runtime_error("memset failed");
        const GasWindowTables T{ngpt, nmax, ncmax};
        int* tbl = gas_window_tables(st, gpoint_flavor, ngpt, nminorlower, nminorupper, ncmax, NXW);
        gas_window_tables_kernel<<<1, 256, size_t(T.ints() + ngpt + 4 + 2*((ngpt + 63)/64))*sizeof(int), st>>>(
                ngpt, nminorlower, nminorupper, ncmax, NXW, gpoint_flavor, minor_limits_gpt_lower, minor_limits_gpt_upper,
                minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper,
                idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper,
                kminor_start_lower, kminor_start_upper, tbl);
        const PlanckArgs<F> pa{pfracin, tlev, tsfc, sfc_lay, nPlanckTemp, gpoint_bands, totplnk_delta, totplnk, pfrac, blay, blev, sfc_src, sfc_src_jac};
#define RRX_GW_PF_ARGS ncol, nlay, ngpt, neta, npres, ntemp, nminorlower, nminorupper, idx_h2o, gpoint_flavor, \
                kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
                minor_scales_with_density_lower, minor_scales_with_density_upper, \
                scale_by_complement_lower, scale_by_complement_upper, \
                idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, \
                kminor_start_lower, kminor_start_upper, play, tlay, col_gas, (const F*)nullptr, (const F*)nullptr, ia, \
                tau, (F*)nullptr, (F*)nullptr, pa, todo, geom, tbl, ncmax
        if (ia.cld_tau != nullptr) gas_window_kernel<F,2,true,true><<<dim3(wgrid.x, wgrid.y, nz), block, wlds, st>>>(RRX_GW_PF_ARGS);
        else gas_window_kernel<F,2,true><<<dim3(wgrid.x, wgrid.y, nz), block, wlds, st>>>(RRX_GW_PF_ARGS);
#undef RRX_GW_PF_ARGS
        gas_window_stats("lw + fractions", todo, nblk*nz, st);
    }
    const dim3 g2 = windowed ? gather_grid(nblk*nz) : grid;
#define RRX_TA_PF_ARGS ncol, nlay, ngpt, neta, npres, ntemp, nminorlower, nminorupper, idx_h2o, gpoint_flavor, \
            kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
            minor_scales_with_density_lower, minor_scales_with_density_upper, \
            scale_by_complement_lower, scale_by_complement_upper, \
            idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, \
            kminor_start_lower, kminor_start_upper, \
            (const Bool*)nullptr, (const F*)nullptr, (const F*)nullptr, (const F*)nullptr, play, tlay, col_gas, (const F*)nullptr, \
            (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, (const F*)nullptr, \
            tau, (F*)nullptr, (F*)nullptr, ia, todo, int(wgrid.x), nblk, nz, windowed ? geom : 0
    if (ia.cld_tau != nullptr) tau_absorption_kernel<F,2,true,true><<<g2, block, lds, st>>>(RRX_TA_PF_ARGS);
    else tau_absorption_kernel<F,2,true><<<g2, block, lds, st>>>(RRX_TA_PF_ARGS);
#undef RRX_TA_PF_ARGS
    planck_fraction_kernel<F><<<g2, block, size_t(2)*ngpt*sizeof(int), st>>>(
            ncol, nlay, ngpt, neta, npres, ntemp, nPlanckTemp, play, tlay, tlev, tsfc, sfc_lay, col_gas, ia, gpoint_bands, pfracin,
            totplnk_delta, totplnk, gpoint_flavor, pfrac, blay, blev, sfc_src, sfc_src_jac, todo, int(wgrid.x), nblk, nz, windowed ? geom : 0);
    RRX_CATCH("rrx_gas_optics_lw_fractions")
}

template<typename F, int MODE, bool DIRECT = false>
int tau_absorption_impl(
        int ncol, int nlay, int nband, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp,
        int nminorlower, int nminorklower, int nminorupper, int nminorkupper, int idx_h2o,
        const int* gpoint_flavor, const int* band_lims_gpt,
        const F* kmajor, const F* kminor_lower, const F* kminor_upper,
        const int* minor_limits_gpt_lower, const int* minor_limits_gpt_upper,
        const Bool* minor_scales_with_density_lower, const Bool* minor_scales_with_density_upper,
        const Bool* scale_by_complement_lower, const Bool* scale_by_complement_upper,
        const int* idx_minor_lower, const int* idx_minor_upper,
        const int* idx_minor_scaling_lower, const int* idx_minor_scaling_upper,
        const int* kminor_start_lower, const int* kminor_start_upper,
        const Bool* tropo, const F* col_mix, const F* fmajor, const F* fminor,
        const F* play, const F* tlay, const F* col_gas, const F* col_dry,
        const int* jeta, const int* jtemp, const int* jpress, const F* krayl,
        F* tau, F* ssa, F* g, void* stream, const char* name, const InterpArgs<F> ia = InterpArgs<F>())
{
    RRX_TRY
    (void)ngas; (void)nflav; (void)band_lims_gpt; (void)nminorklower; (void)nminorkupper;
    if (ncol <= 0 || nlay <= 0 || ngpt <= 0) throw std::runtime_error("empty problem");
    const int nchunk = (ngpt + GCH - 1) / GCH;
    const int nmax = std::max(nminorlower, nminorupper);
    const size_t lds = (size_t(3)*ngpt + size_t(2)*nchunk*(1 + ITEM*nmax) + size_t(2)*MM*nmax)*sizeof(int);
    if (lds > 64*1024) throw std::runtime_error("minor-gas index exceeds 64 KiB of LDS");
    const dim3 block(64, 4);
    const dim3 grid(ceil_div(ncol, 64), ceil_div(nlay, 4));
    if constexpr (DIRECT && MODE != 0)
    {
        // windowed kernel first; the gather kernel then finishes the workgroups it handed back (usually none)
        const int ncmax = gas_window_ncmax(ngpt, nband);
        const size_t wlds = gas_window_lds_bytes<F>(ngpt, nmax, ncmax, MODE, false);
        // (32-bit byte offsets inside a g-point slab and inside the kmajor table, see gas_optics_lw_fractions_impl)
        if (tuning().go_window && wlds <= 64*1024 && size_t(ncol)*nlay*sizeof(F) < (size_t(1) << 32)
            && size_t(ngpt)*ntemp*neta*(npres+1)*sizeof(F) < (size_t(1) << 32))
        {
            hipStream_t st = static_cast<hipStream_t>(stream);
            StreamScratch scratch(st);
            const int geom = gas_window_geometry(ncol);
            const dim3 wgrid = gas_window_grid(geom, ncol, nlay);
            const int nblk = int(wgrid.x)*int(wgrid.y);
            const int nz = gas_window_parts(nblk, nchunk);
            int* todo = scratch.get<int>(size_t(9) + size_t(nblk)*nz) + 8;
            if (hipMemsetAsync(todo - 8, 0, 9*sizeof(int), st) != hipSuccess) throw std::runtime_error("memset failed");
            const GasWindowTables T{ngpt, nmax, ncmax};
            int* tbl = gas_window_tables(st, gpoint_flavor, ngpt, nminorlower, nminorupper, ncmax, (MODE == 1) ? NCW : NXW);
            gas_window_tables_kernel<<<1, 256, size_t(T.ints() + ngpt + 4 + 2*((ngpt + 63)/64))*sizeof(int), st>>>(
                    ngpt, nminorlower, nminorupper, ncmax, (MODE == 1) ? NCW : NXW, gpoint_flavor, minor_limits_gpt_lower, minor_limits_gpt_upper,
                    minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper,
                    idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper,
                    kminor_start_lower, kminor_start_upper, tbl);
#define RRX_GW_ARGS ncol, nlay, ngpt, neta, npres, ntemp, nminorlower, nminorupper, idx_h2o, gpoint_flavor, \
                    kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
                    minor_scales_with_density_lower, minor_scales_with_density_upper, \
                    scale_by_complement_lower, scale_by_complement_upper, \
                    idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, \
                    kminor_start_lower, kminor_start_upper, play, tlay, col_gas, col_dry, krayl, ia, tau, ssa, g, \
                    PlanckArgs<F>(), todo, geom, tbl, ncmax
#define RRX_TA_ARGS ncol, nlay, ngpt, neta, npres, ntemp, nminorlower, nminorupper, idx_h2o, gpoint_flavor, \
                    kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
                    minor_scales_with_density_lower, minor_scales_with_density_upper, \
                    scale_by_complement_lower, scale_by_complement_upper, \
                    idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, \
                    kminor_start_lower, kminor_start_upper, \
                    tropo, col_mix, fmajor, fminor, play, tlay, col_gas, col_dry, jeta, jtemp, jpress, krayl, \
                    tau, ssa, g, ia, todo, int(wgrid.x), nblk, nz, geom
            const bool cld = DIRECT && MODE != 0 && ia.cld_tau != nullptr;
            if constexpr (DIRECT && MODE != 0)
            {
                if (cld) gas_window_kernel<F,MODE,false,true><<<dim3(wgrid.x, wgrid.y, nz), block, wlds, st>>>(RRX_GW_ARGS);
            }
            if (!cld) gas_window_kernel<F,MODE,false><<<dim3(wgrid.x, wgrid.y, nz), block, wlds, st>>>(RRX_GW_ARGS);
            gas_window_stats(MODE == 1 ? "sw" : "lw", todo, nblk*nz, st);
            if constexpr (DIRECT && MODE != 0)
            {
                if (cld) tau_absorption_kernel<F,MODE,DIRECT,true><<<gather_grid(nblk*nz), block, lds, st>>>(RRX_TA_ARGS);
            }
            if (!cld) tau_absorption_kernel<F,MODE,DIRECT><<<gather_grid(nblk*nz), block, lds, st>>>(RRX_TA_ARGS);
#undef RRX_GW_ARGS
#undef RRX_TA_ARGS
            return check_launch(name);
        }
    }
#define RRX_TA_ARGS ncol, nlay, ngpt, neta, npres, ntemp, nminorlower, nminorupper, idx_h2o, gpoint_flavor, \
            kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
            minor_scales_with_density_lower, minor_scales_with_density_upper, \
            scale_by_complement_lower, scale_by_complement_upper, \
            idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, \
            kminor_start_lower, kminor_start_upper, \
            tropo, col_mix, fmajor, fminor, play, tlay, col_gas, col_dry, jeta, jtemp, jpress, krayl, \
            tau, ssa, g, ia
    if constexpr (DIRECT && MODE != 0)
    {
        if (ia.cld_tau != nullptr)
        {
            tau_absorption_kernel<F,MODE,DIRECT,true><<<grid, block, lds, static_cast<hipStream_t>(stream)>>>(RRX_TA_ARGS);
            return check_launch(name);
        }
    }
    tau_absorption_kernel<F,MODE,DIRECT><<<grid, block, lds, static_cast<hipStream_t>(stream)>>>(RRX_TA_ARGS);
#undef RRX_TA_ARGS
    RRX_CATCH(name)
}
}  // namespace


extern "C"
{
#define RRX_DEFINE_GAS(F, SFX) \
int rrx_interpolation##SFX( \
        int ncol, int nlay, int ngas, int nflav, int neta, int npres, int ntemp, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, \
        const F* vmr_ref, const F* play, const F* tlay, F* col_gas, \
        int* jtemp, F* fmajor, F* fminor, F* col_mix, RrxBool* tropo, int* jeta, int* jpress, void* stream) \
{ \
    RRX_TRY \
    if (ncol <= 0 || nlay <= 0) throw std::runtime_error("empty problem"); \
    interpolation_kernel<F><<<dim3(rrx::ceil_div(ncol, 64), rrx::ceil_div(nlay, 4)), dim3(64, 4), 0, static_cast<hipStream_t>(stream)>>>( \
            ncol, nlay, ngas, nflav, neta, npres, ntemp, flavor, press_ref_log, temp_ref, \
            press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log, vmr_ref, play, tlay, col_gas, \
            jtemp, fmajor, fminor, col_mix, tropo, jeta, jpress); \
    RRX_CATCH("rrx_interpolation") \
} \
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
        const int* jeta, const int* jtemp, const int* jpress, F* tau, void* stream) \
{ \
    return tau_absorption_impl<F,0>(ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp, \
            nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt, \
            kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
            minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper, \
            idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, \
            tropo, col_mix, fmajor, fminor, play, tlay, col_gas, (const F*)nullptr, jeta, jtemp, jpress, (const F*)nullptr, \
            tau, (F*)nullptr, (F*)nullptr, stream, "rrx_compute_tau_absorption"); \
} \
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
        const int* jeta, const int* jtemp, const int* jpress, F* tau, void* stream) \
{ \
    return tau_absorption_impl<F,2>(ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp, \
            nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt, \
            kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
            minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper, \
            idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, \
            tropo, col_mix, fmajor, fminor, play, tlay, col_gas, (const F*)nullptr, jeta, jtemp, jpress, (const F*)nullptr, \
            tau, (F*)nullptr, (F*)nullptr, stream, "rrx_compute_tau_absorption_set"); \
} \
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
        F* tau, F* ssa, F* g, void* stream) \
{ \
    return tau_absorption_impl<F,1>(ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp, \
            nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt, \
            kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
            minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper, \
            idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, \
            tropo, col_mix, fmajor, fminor, play, tlay, col_gas, col_dry, jeta, jtemp, jpress, krayl, \
            tau, ssa, g, stream, "rrx_gas_optics_sw_fused"); \
} \
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
        const F* play, const F* tlay, const F* col_gas, F* tau, void* stream) \
{ \
    const InterpArgs<F> ia{ngas, flavor, press_ref_log, temp_ref, press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log, vmr_ref}; \
    return tau_absorption_impl<F,2,true>(ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp, \
            nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt, \
            kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
            minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper, \
            idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, \
            (const RrxBool*)nullptr, (const F*)nullptr, (const F*)nullptr, (const F*)nullptr, play, tlay, col_gas, (const F*)nullptr, \
            (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, (const F*)nullptr, \
            tau, (F*)nullptr, (F*)nullptr, stream, "rrx_gas_optics_lw_direct", ia); \
} \
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
        const F* play, const F* tlay, const F* col_gas, F* tau, const F* cld_tau, void* stream) \
{ \
    const InterpArgs<F> ia{ngas, flavor, press_ref_log, temp_ref, press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log, vmr_ref, \
            cld_tau, (const F*)nullptr, (const F*)nullptr, band_lims_gpt}; \
    return tau_absorption_impl<F,2,true>(ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp, \
            nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt, \
            kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
            minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper, \
            idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, \
            (const RrxBool*)nullptr, (const F*)nullptr, (const F*)nullptr, (const F*)nullptr, play, tlay, col_gas, (const F*)nullptr, \
            (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, (const F*)nullptr, \
            tau, (F*)nullptr, (F*)nullptr, stream, "rrx_gas_optics_lw_direct_allsky", ia); \
} \
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
        F* tau, F* ssa, F* g, void* stream) \
{ \
    const InterpArgs<F> ia{ngas, flavor, press_ref_log, temp_ref, press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log, vmr_ref}; \
    return tau_absorption_impl<F,1,true>(ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp, \
            nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt, \
            kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
            minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper, \
            idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, \
            (const RrxBool*)nullptr, (const F*)nullptr, (const F*)nullptr, (const F*)nullptr, play, tlay, col_gas, col_dry, \
            (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, krayl, \
            tau, ssa, g, stream, "rrx_gas_optics_sw_direct", ia); \
} \
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
        F* tau, F* ssa, F* g, const F* cld_tau, const F* cld_ssa, const F* cld_g, void* stream) \
{ \
    if (cld_tau != nullptr && (cld_ssa == nullptr || cld_g == nullptr || g == nullptr)) { rrx::set_error("rrx_gas_optics_sw_direct_allsky: by-band ssa, g and the g output are needed with by-band tau"); return 1; } \
    const InterpArgs<F> ia{ngas, flavor, press_ref_log, temp_ref, press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log, vmr_ref, \
            cld_tau, cld_ssa, cld_g, band_lims_gpt}; \
    return tau_absorption_impl<F,1,true>(ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp, \
            nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, gpoint_flavor, band_lims_gpt, \
            kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
            minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper, \
            idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, \
            (const RrxBool*)nullptr, (const F*)nullptr, (const F*)nullptr, (const F*)nullptr, play, tlay, col_gas, col_dry, \
            (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, krayl, \
            tau, ssa, g, stream, "rrx_gas_optics_sw_direct_allsky", ia); \
} \
int rrx_planck_source_direct##SFX( \
        int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, int nPlanckTemp, \
        const F* play, const F* tlay, const F* tlev, const F* tsfc, int sfc_lay, const F* col_gas, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, const F* vmr_ref, \
        const int* gpoint_bands, const int* band_lims_gpt, const F* pfracin, \
        F totplnk_delta, const F* totplnk, const int* gpoint_flavor, \
        F* sfc_src, F* lay_src, F* lev_src, F* sfc_src_jac, void* stream) \
{ \
    RRX_TRY \
    (void)nbnd; (void)nflav; (void)band_lims_gpt; \
    if (ncol <= 0 || nlay <= 0 || ngpt <= 0) throw std::runtime_error("empty problem"); \
    const InterpArgs<F> ia{ngas, flavor, press_ref_log, temp_ref, press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log, vmr_ref}; \
    planck_source_kernel<F,true><<<dim3(rrx::ceil_div(ncol, 64), rrx::ceil_div(nlay, PL)), dim3(64, PL), planck_lds_bytes<F>(ngpt), static_cast<hipStream_t>(stream)>>>( \
            ncol, nlay, ngpt, neta, npres, ntemp, nPlanckTemp, tlay, tlev, tsfc, sfc_lay, (const F*)nullptr, (const int*)nullptr, (const RrxBool*)nullptr, \
            (const int*)nullptr, (const int*)nullptr, gpoint_bands, pfracin, temp_ref_min, totplnk_delta, totplnk, gpoint_flavor, \
            sfc_src, lay_src, lev_src, sfc_src_jac, tuning().go_share, play, col_gas, ia); \
    RRX_CATCH("rrx_planck_source_direct") \
} \
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
        F* tau, F* pfrac, F* blay, F* blev, F* sfc_src, F* sfc_src_jac, void* stream) \
{ \
    (void)nminorklower; (void)nminorkupper; (void)band_lims_gpt; \
    const InterpArgs<F> ia{ngas, flavor, press_ref_log, temp_ref, press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log, vmr_ref}; \
    return gas_optics_lw_fractions_impl<F>(ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp, nPlanckTemp, nminorlower, nminorupper, idx_h2o, \
            gpoint_flavor, gpoint_bands, kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
            minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper, \
            idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, \
            ia, play, tlay, tlev, tsfc, sfc_lay, col_gas, pfracin, totplnk_delta, totplnk, tau, pfrac, blay, blev, sfc_src, sfc_src_jac, stream); \
} \
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
        F* tau, F* pfrac, F* blay, F* blev, F* sfc_src, F* sfc_src_jac, const F* cld_tau, void* stream) \
{ \
    (void)nminorklower; (void)nminorkupper; \
    const InterpArgs<F> ia{ngas, flavor, press_ref_log, temp_ref, press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log, vmr_ref, \
            cld_tau, (const F*)nullptr, (const F*)nullptr, band_lims_gpt}; \
    return gas_optics_lw_fractions_impl<F>(ncol, nlay, nband, ngpt, ngas, nflav, neta, npres, ntemp, nPlanckTemp, nminorlower, nminorupper, idx_h2o, \
            gpoint_flavor, gpoint_bands, kmajor, kminor_lower, kminor_upper, minor_limits_gpt_lower, minor_limits_gpt_upper, \
            minor_scales_with_density_lower, minor_scales_with_density_upper, scale_by_complement_lower, scale_by_complement_upper, \
            idx_minor_lower, idx_minor_upper, idx_minor_scaling_lower, idx_minor_scaling_upper, kminor_start_lower, kminor_start_upper, \
            ia, play, tlay, tlev, tsfc, sfc_lay, col_gas, pfracin, totplnk_delta, totplnk, tau, pfrac, blay, blev, sfc_src, sfc_src_jac, stream); \
} \
int rrx_planck_fractions##SFX( \
        int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, int nPlanckTemp, \
        const F* play, const F* tlay, const F* tlev, const F* tsfc, int sfc_lay, const F* col_gas, \
        const int* flavor, const F* press_ref_log, const F* temp_ref, \
        F press_ref_log_delta, F temp_ref_min, F temp_ref_delta, F press_ref_trop_log, const F* vmr_ref, \
        const int* gpoint_bands, const int* band_lims_gpt, const F* pfracin, \
        F totplnk_delta, const F* totplnk, const int* gpoint_flavor, \
        F* pfrac, F* blay, F* blev, F* sfc_src, F* sfc_src_jac, void* stream) \
{ \
    RRX_TRY \
    (void)nbnd; (void)nflav; (void)band_lims_gpt; \
    if (ncol <= 0 || nlay <= 0 || ngpt <= 0) throw std::runtime_error("empty problem"); \
    const InterpArgs<F> ia{ngas, flavor, press_ref_log, temp_ref, press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log, vmr_ref}; \
    planck_fraction_kernel<F><<<dim3(rrx::ceil_div(ncol, 64), rrx::ceil_div(nlay, 4)), dim3(64, 4), size_t(2)*ngpt*sizeof(int), static_cast<hipStream_t>(stream)>>>( \
            ncol, nlay, ngpt, neta, npres, ntemp, nPlanckTemp, play, tlay, tlev, tsfc, sfc_lay, col_gas, ia, gpoint_bands, pfracin, \
            totplnk_delta, totplnk, gpoint_flavor, pfrac, blay, blev, sfc_src, sfc_src_jac); \
    RRX_CATCH("rrx_planck_fractions") \
} \
int rrx_compute_tau_rayleigh##SFX( \
        int ncol, int nlay, int nbnd, int ngpt, int ngas, int nflav, int neta, int npres, int ntemp, \
        const int* gpoint_flavor, const int* band_lims_gpt, const F* krayl, \
        int idx_h2o, const F* col_dry, const F* col_gas, \
        const F* fminor, const int* jeta, const RrxBool* tropo, const int* jtemp, F* tau_rayleigh, void* stream) \
{ \
    RRX_TRY \
    (void)nbnd; (void)ngas; (void)nflav; (void)npres; (void)band_lims_gpt; \
    if (ncol <= 0 || nlay <= 0 || ngpt <= 0) throw std::runtime_error("empty problem"); \
    tau_rayleigh_kernel<F><<<dim3(rrx::ceil_div(ncol, 64), rrx::ceil_div(nlay, 4)), dim3(64, 4), 0, static_cast<hipStream_t>(stream)>>>( \
            ncol, nlay, ngpt, neta, ntemp, gpoint_flavor, krayl, idx_h2o, col_dry, col_gas, fminor, jeta, tropo, jtemp, tau_rayleigh); \
    RRX_CATCH("rrx_compute_tau_rayleigh") \
} \
int rrx_combine_abs_and_rayleigh##SFX(int ncol, int nlay, int ngpt, const F* tau_abs, const F* tau_rayleigh, F* tau, F* ssa, F* g, void* stream) \
{ \
    RRX_TRY \
    const size_t n = size_t(ncol)*nlay*ngpt; \
    combine_kernel<F><<<grid1d(n), 256, 0, static_cast<hipStream_t>(stream)>>>(n, tau_abs, tau_rayleigh, tau, ssa, g); \
    RRX_CATCH("rrx_combine_abs_and_rayleigh") \
} \
int rrx_compute_planck_source##SFX( \
        int ncol, int nlay, int nbnd, int ngpt, int nflav, int neta, int npres, int ntemp, int nPlanckTemp, \
        const F* tlay, const F* tlev, const F* tsfc, int sfc_lay, \
        const F* fmajor, const int* jeta, const RrxBool* tropo, const int* jtemp, const int* jpress, \
        const int* gpoint_bands, const int* band_lims_gpt, const F* pfracin, \
        F temp_ref_min, F totplnk_delta, const F* totplnk, const int* gpoint_flavor, \
        F* sfc_src, F* lay_src, F* lev_src, F* sfc_src_jac, void* stream) \
{ \
    RRX_TRY \
    (void)nbnd; (void)nflav; (void)band_lims_gpt; \
    if (ncol <= 0 || nlay <= 0 || ngpt <= 0) throw std::runtime_error("empty problem"); \
    planck_source_kernel<F><<<dim3(rrx::ceil_div(ncol, 64), rrx::ceil_div(nlay, PL)), dim3(64, PL), planck_lds_bytes<F>(ngpt), static_cast<hipStream_t>(stream)>>>( \
            ncol, nlay, ngpt, neta, npres, ntemp, nPlanckTemp, tlay, tlev, tsfc, sfc_lay, fmajor, jeta, tropo, jtemp, jpress, \
            gpoint_bands, pfracin, temp_ref_min, totplnk_delta, totplnk, gpoint_flavor, sfc_src, lay_src, lev_src, sfc_src_jac, tuning().go_share, \
            (const F*)nullptr, (const F*)nullptr, InterpArgs<F>()); \
    RRX_CATCH("rrx_compute_planck_source") \
} \
int rrx_reorder123x321##SFX(int ni, int nj, int nk, const F* arr_in, F* arr_out, void* stream) \
{ \
    RRX_TRY \
    reorder123x321_kernel<F><<<grid1d(size_t(ni)*nj*nk), 256, 0, static_cast<hipStream_t>(stream)>>>(ni, nj, nk, arr_in, arr_out); \
    RRX_CATCH("rrx_reorder123x321") \
} \
int rrx_reorder12x21##SFX(int ni, int nj, const F* arr_in, F* arr_out, void* stream) \
{ \
    RRX_TRY \
    reorder12x21_kernel<F><<<grid1d(size_t(ni)*nj), 256, 0, static_cast<hipStream_t>(stream)>>>(ni, nj, arr_in, arr_out); \
    RRX_CATCH("rrx_reorder12x21") \
} \
int rrx_zero_array##SFX(int ni, int nj, int nk, F* arr, void* stream) \
{ \
    RRX_TRY \
    if (hipMemsetAsync(arr, 0, size_t(ni)*nj*nk*sizeof(F), static_cast<hipStream_t>(stream)) != hipSuccess) \
        throw std::runtime_error("hipMemsetAsync failed"); \
    RRX_CATCH("rrx_zero_array") \
}

RRX_DEFINE_GAS(double, _f64)
RRX_DEFINE_GAS(float, _f32)
}

extern "C" int rrx_gas_window_stats(long long* handed_back, long long* workgroups, int reset)
{
    if (handed_back) *handed_back = g_gw_handed;
    if (workgroups) *workgroups = g_gw_total;
    if (reset) { g_gw_handed = 0; g_gw_total = 0; }
    return 0;
}
