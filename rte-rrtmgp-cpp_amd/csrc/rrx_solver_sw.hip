// Shortwave two-stream solver, fused: boundary conditions + two-stream coefficients + direct-beam source +
// adding (albedo/source upward, fluxes downward) in ONE kernel with no global scratch. Replaces
//   Rte_solver_kernels_cuda::sw_solver_2stream  (/root/reference/src_kernels_cuda/rte_solver_kernels_launchers.cu:289-447)
// and its kernels apply_BC_kernel (x3), sw_source_2stream_kernel + sw_2stream_function, sw_adding_kernel
// (/root/reference/src_kernels_cuda/rte_solver_kernels.cu:196-286,351-387,543-655).
//
// MI355X design (DESIGN.md section "sw_solver_2stream"): same wave tiling as the LW solver -- 8 column-lanes x
// 8 level-lanes, each level-lane holding K consecutive layers in registers, every input read once and every
// output written once. The four vertical recurrences are propagated between level-lanes with 3-step __shfl scans:
//   direct beam      dir' = t_noscat * dir                                  (product scan, downward)
//   albedo           a'   = r + t^2 a / (1 - r a)   = Moebius map [[t^2-r^2, r], [-r, 1]]   (2x2 matrix scan, upward)
//   diffuse source   s'   = alpha s + beta          (affine scan, upward;  alpha = t/(1 - r a))
//   diffuse down     d'   = alpha d + b             (affine scan, downward)
// Inside a lane the K layers are always replayed with the reference's own formulas.
#include "rrx_common.h"
#include "rrx_hip.h"
#include <type_traits>

#pragma clang fp contract(fast)

namespace
{
using namespace rrx;

constexpr int CL = 8;
constexpr int LL = 8;
constexpr int LOADG = 6;   // layers per load group
#ifndef RRX_SW_BB_LOADS
#define RRX_SW_BB_LOADS 3
#endif
#ifndef RRX_SW_BB_EVALS
#define RRX_SW_BB_EVALS 2
#endif
constexpr int BB_LOADS = RRX_SW_BB_LOADS;   // fused broadband form: layers of loads in flight ahead of the evaluation
constexpr int BB_EVALS = RRX_SW_BB_EVALS;   // fused broadband form: two_stream evaluations the scheduler may interleave
#ifndef RRX_SW_MINWAVES
#define RRX_SW_MINWAVES 1
#endif
#ifndef RRX_SW_MINWAVES2
#define RRX_SW_MINWAVES2 2
#endif
#ifndef RRX_SW_XCD_MAP
#define RRX_SW_XCD_MAP 1
#endif
#ifndef RRX_SW_F32_NW
#define RRX_SW_F32_NW 4        // wavefronts per workgroup of the fp32 geometry: 4 = ONE column group (below), 8 = two (A/B)
#endif
#ifndef RRX_SW_F32_WAVES1
#define RRX_SW_F32_WAVES1 3    // ... and the waves per SIMD the one-group form is compiled for
#endif
#ifndef RRX_SW_F32_WAVES
#define RRX_SW_F32_WAVES 2    // waves per SIMD the fp32 geometry (16 x 4 lanes, one column per lane) is compiled for
#endif

#ifndef RRX_SW_TIMING
#define RRX_SW_TIMING 0   // diagnostic build (tools/sw_timing.sh): wavefront 0 of every workgroup adds the clocks it spends per phase of a g-point to g_sw_clk
#endif
#if RRX_SW_TIMING
__device__ unsigned long long g_sw_clk[16][8];      // [wavefront of the workgroup][phase]
#define RRX_SW_T(k) { const unsigned long long t_ = __builtin_readcyclecounter(); sw_acc[k] += t_ - sw_t; sw_t = t_; }
#else
#define RRX_SW_T(k)
#endif

template<typename F>
struct TwoStream { F r_dif, t_dif, r_dir, t_dir, t_noscat; };

// /root/reference/src_kernels_cuda/rte_solver_kernels.cu:543-592 (Zdunkowski PIFM two-stream, Ukkonen clamps).
// Same formulas; the three divisions per cell (1/mu0, rt_term, /fact) are replaced by one hoisted reciprocal of mu0
// and ONE Newton reciprocal x = 1/(D*fact): rt_term = x*fact, rt_term2 = ssa*x (fp64 vector rate is the scarce
// resource of this kernel: DESIGN.md section "sw_solver_2stream"). fp64: exp and sqrt are the lean forms of
// rrx_common.h (arguments are <= 0 resp. in [1e-12, 16]). GZ: asymmetry identically zero (clear-sky gas optics), the
// same expressions with g = 0 folded in by hand: gamma3 = gamma4 = 1/2, alpha1 = alpha2 = (gamma1 + gamma2)/2.
template<typename F, bool GZ = false>
__device__ __forceinline__ TwoStream<F> two_stream(const F tau, const F ssa, const F g, const F mu0, const F mu0_inv)
{
    TwoStream<F> o;
    const F tmin = Lim<F>::eps();
    F gamma1, gamma2, alpha1, alpha2, k_gamma3, k_gamma4, k;
    if constexpr (GZ)
    {
        gamma1 = fma(F(-1.25), ssa, F(2.));
        gamma2 = F(.75) * ssa;
        const F sum = gamma1 + gamma2;
        k = sqrt_pos(max((gamma1 - gamma2) * sum, Lim<F>::k_min()));
        alpha1 = alpha2 = F(.5) * sum;
        k_gamma3 = k_gamma4 = F(.5) * k;
    }
    else
    {
        gamma1 = (F(8.) - ssa * (F(5.) + F(3.) * g)) * F(.25);
        gamma2 = F(3.) * (ssa * (F(1.) - g)) * F(.25);
        const F gamma3 = (F(2.) - F(3.) * mu0 * g) * F(.25);
        const F gamma4 = F(1.) - gamma3;
        alpha1 = gamma1 * gamma4 + gamma2 * gamma3;
        alpha2 = gamma1 * gamma3 + gamma2 * gamma4;
        k = sqrt_pos(max((gamma1 - gamma2) * (gamma1 + gamma2), Lim<F>::k_min()));
        k_gamma3 = k * gamma3;
        k_gamma4 = k * gamma4;
    }
    const F exp_minusktau = exp_neg(-tau * k);
    const F exp_minus2ktau = exp_minusktau * exp_minusktau;
    const F k_mu = k * mu0;
    const F omk2 = F(1.) - k_mu*k_mu;
    const F fact = (abs(omk2) > tmin) ? omk2 : tmin;
    const F D = k * (F(1.) + exp_minus2ktau) + gamma1 * (F(1.) - exp_minus2ktau);
    const F x = fast_rcp(D * fact);
    const F rt_term = x * fact, rt_term2 = ssa * x;
    o.r_dif = rt_term * gamma2 * (F(1.) - exp_minus2ktau);
    o.t_dif = rt_term * F(2.) * k * exp_minusktau;
    o.t_noscat = exp_neg(-tau * mu0_inv);
    const F r_dir = rt_term2 * ((F(1.) - k_mu) * (alpha2 + k_gamma3) -
                                (F(1.) + k_mu) * (alpha2 - k_gamma3) * exp_minus2ktau -
                                F(2.) * (k_gamma3 - alpha2 * k_mu) * exp_minusktau * o.t_noscat);
    const F t_dir = -rt_term2 * ((F(1.) + k_mu) * (alpha1 + k_gamma4) * o.t_noscat -
                                 (F(1.) - k_mu) * (alpha1 - k_gamma4) * exp_minus2ktau * o.t_noscat -
                                 F(2.) * (k_gamma4 + alpha1 * k_mu) * exp_minusktau);
    o.r_dir = max(tmin, min(r_dir, F(1.) - o.t_noscat));
    o.t_dir = max(tmin, min(t_dir, F(1.) - o.t_noscat - o.r_dir));
    return o;
}


// W = waves per column group: W = 1 keeps the whole column in one wavefront (8 level-lanes); W = 2 spreads the levels of
// 8*V columns over 16 level-lanes in two adjacent wavefronts (half the per-lane state, so more resident waves per
// SIMD); the four vertical scans then exchange each wave's total through LDS with one block barrier per scan.
// BB (broadband): the workgroup walks over ALL g-points of its columns and keeps the g-point sums of the three fluxes
// on chip (up and dn in LDS, dir in registers), added in g-point order like sum_broadband does on stored per-g-point
// fluxes, so the same bits; flux_up/dn/dir are then (ncol, nlev) arrays.
// PRE (BB form without g array): software pipeline over the g-point loop. The K layers of tau and ssa of g-point g+1 are
// requested right after the two-stream phase of g-point g, when the registers of its temporaries are free, and land
// during the scans and replays; the next iteration finds them in registers. Measured at C4 (tools/labs/sw_lab.hip): 5.3 -> 4.7 ms;
// loads issued layer by layer inside the two-stream phase were still in flight when their layer came up.
// W = 4 (round 3): four wavefronts per column group, eight per workgroup (two column groups, as with W = 2): columns of up to 287
// layers at nine layers per lane; the wave totals of a scan are then combined over the group's waves in order (the W = 2 code is
// the two-wave case of the same composition, kept as it was).
// CLT = column lanes per wavefront (level lanes = 64 / CLT). 8 x 8 is the fp64 geometry (64-B rows per wave, two column groups per
// workgroup share each 128-B line). Round 4, fp32: 16 x 4 lanes with ONE column per lane, four waves per column group and two groups
// per workgroup -- the same nine cells per lane and the same 64-B rows as fp64, no scratch (the two-columns-per-lane form of rounds
// 1-3 spilled 25-34 VGPRs): 3.70 -> 3.33 ms at C4 clear sky, 8.06 -> 7.5 ms all-sky at 32 768 columns. What was measured on the
// way (profiles/r04_issue_costs_fp32.txt, r04_fp32_geometry_ab.txt): packed v_pk_fma_f32 issues every 3.8 cycles against 2.0 for
// v_fma_f32 at four waves per SIMD (5.7 against 3.5 at two), so two columns per lane on packed math buy at most a fifth of the
// packable instructions; compiled for four waves per SIMD (128 VGPRs) this geometry spills 36-65 registers and is slower (4.44 /
// 12.8 ms); K = 6 with six waves per column group (three waves per SIMD, no spills) 3.43 / 7.95 ms and K = 5 with eight (four
// per SIMD) 3.61 / 8.6 ms: the wider exchanges cost what the occupancy brings.
// Late round 4: ONE column group per workgroup (NW = W = 4, 256 threads, half the LDS) compiled for THREE waves per SIMD (<= 168
// VGPRs: 150-168 without scratch) -- three workgroups per CU instead of one, and an fp32 instruction issues every 2.5 cycles at
// three waves per SIMD against 3.8 at two (tools/issue_mix_bench.hip): 3.11 -> 2.24 ms at C4, 7.07 -> 5.53 ms all-sky at 32 768
// columns. The 64-B rows of a group are then half a 128-B line whose other half belongs to the next workgroup (L2 serves it);
// compiled for four waves per SIMD the same form spills (2.65 / 9.3 ms).
template<typename F, int V, int K, int W, bool BB = false, bool GZ = false, bool PRE = false, bool GS = false, int NW = (W > 2 ? 2*W : 4), int CLT = 8>
__global__ void __launch_bounds__(64*NW, (CLT == 16) ? ((NW == 4) ? RRX_SW_F32_WAVES1 : RRX_SW_F32_WAVES) : ((NW > 4) ? 1 : ((W == 2 && V*sizeof(F) <= 8) ? RRX_SW_MINWAVES2 : RRX_SW_MINWAVES)))
sw_2stream_scan_kernel(
        const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* __restrict__ tau, const F* __restrict__ ssa, const F* __restrict__ g, const F* __restrict__ mu0,
        const F* __restrict__ sfc_alb_dir, const F* __restrict__ sfc_alb_dif,
        const F* __restrict__ inc_flux_dir, const F* __restrict__ inc_flux_dif,
        F* __restrict__ flux_up, F* __restrict__ flux_dn, F* __restrict__ flux_dir, const int sync_waves, const int gper)
{
    constexpr int CL = CLT, LL = 64/CLT;              // shadow the default geometry
    // per-thread private LDS columns (dynamic register indexing is not needed: j is a compile-time constant, but
    // two of the six per-layer arrays live here so that the kernel fits 2 waves per SIMD)
    __shared__ F lds_alb[K*V][64*NW];
    __shared__ F lds_dir[K*V][64*NW];
    __shared__ F xch[(W >= 2) ? 8*V : 1][NW][CL];   // wave totals of the scans (one slot per scan component)
    __shared__ F lds_acc_up[BB ? K*V : 1][64*NW];
    __shared__ F lds_acc_dn[BB ? K*V : 1][64*NW];

    const int tid = threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int cl = lane & (CL-1);
    const int ll = lane / CL;
    const int h = (W >= 2) ? (wave % W) : 0;          // which part of the column this wave holds (0 = TOA side)
    [[maybe_unused]] const int w0 = wave - h;         // first wave of the column group
    // (a workgroup whose row segment is half a 128-B line: the other half belongs to the next workgroup -- rrx::xcd_contiguous)
    const int bx = (BB && (NW/W)*CL*V*sizeof(F) < 128 && RRX_SW_XCD_MAP) ? xcd_contiguous(blockIdx.x, gridDim.x) : int(blockIdx.x);
    const int wave_col0 = (bx*(NW/W) + wave/W) * (CL*V);
    if constexpr (W == 1) { if (wave_col0 >= ncol) return; }

    // W == 2: every wave stays alive until the last barrier; lanes without a column compute on a clamped one
    int icol = wave_col0 + cl*V;
    const bool active = icol < ncol;
    if (!active) icol = (wave_col0 < ncol) ? wave_col0 : 0;

    const bool writer = active && wave_col0 < ncol;
    const int nlev = nlay + 1;
    const size_t ncl = size_t(ncol);
    const int t0 = (h*LL + ll)*K;

    const Vec<F,V> mu = load_cols<F,V>(mu0 + icol);
    F mu_inv[V];
    #pragma unroll
    for (int v=0; v<V; ++v) mu_inv[v] = F(1.)/mu.v[v];

    F acc_dir[BB ? K : 1][V];
    if constexpr (BB)
    {
        #pragma unroll
        for (int j=0; j<K; ++j)
            #pragma unroll
            for (int v=0; v<V; ++v) { acc_dir[j][v] = F(0.); lds_acc_up[j*V+v][tid] = F(0.); lds_acc_dn[j*V+v][tid] = F(0.); }
    }

    // BB: blockIdx.y = g-point range of this workgroup, its sums go to partial array blockIdx.y (one range: the outputs)
    const int g_begin = BB ? (GS ? blockIdx.y*gper : 0) : blockIdx.y;
    const int g_end = BB ? (GS ? min(ngpt, g_begin + gper) : ngpt) : blockIdx.y + 1;

    // PRE: element offset of layer j inside one g-point slab, recomputed where needed (9 registers less than keeping them)
    auto off_of = [&](const int j) -> unsigned
    {
        const int ml = top_at_1 ? min(t0 + j, nlay-1) : max(nlay-1-t0-j, 0);
        return unsigned(ml)*unsigned(ncol) + unsigned(icol);
    };
    Vec<F,V> nt[PRE ? K : 1], nw[PRE ? K : 1], ng[(PRE && !GZ) ? K : 1], n_inc, n_adir, n_adif;
    if constexpr (PRE)
    {
        static_assert(BB && W >= 2, "the pipelined form is the fused broadband kernel");
        static_assert(GZ || sizeof(F) == 4, "fp64 with a g array: 27 prefetched doubles per lane spill (5.8 -> 7.2 ms, round 3)");
        const F* __restrict__ tau_0 = tau + size_t(g_begin)*ncl*nlay;
        const F* __restrict__ ssa_0 = ssa + size_t(g_begin)*ncl*nlay;
        #pragma unroll
        for (int j=0; j<K; ++j) { const unsigned o = off_of(j); nt[j] = load_cols<F,V>(tau_0 + o); nw[j] = load_cols<F,V>(ssa_0 + o); }
        if constexpr (!GZ)
        {
            const F* __restrict__ g_0 = g + size_t(g_begin)*ncl*nlay;
            #pragma unroll
            for (int j=0; j<K; ++j) ng[j] = load_cols<F,V>(g_0 + off_of(j));
        }
        const size_t s0 = size_t(g_begin)*ncl + icol;
        n_inc = load_cols<F,V>(inc_flux_dir + s0); n_adir = load_cols<F,V>(sfc_alb_dir + s0); n_adif = load_cols<F,V>(sfc_alb_dif + s0);
    }

#if RRX_SW_TIMING
    unsigned long long sw_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sw_t = __builtin_readcyclecounter();
#endif
    for (int igpt=g_begin; igpt<g_end; ++igpt)
    {
    RRX_SW_T(7)
    // partner waves sharing 128-B lines issue their load bursts together (see rrx_solver_lw.hip); the pipelined form
    // issues them behind the first scan barrier instead
    if constexpr (!PRE) { if (sync_waves) __syncthreads(); }
    const size_t lay_base = size_t(igpt)*ncl*nlay + icol;
    const size_t lev_base = size_t(igpt)*ncl*nlev + icol;
    const size_t sfc_idx = size_t(igpt)*ncl + icol;

    // per-layer state; names follow their LAST meaning
    F rp[K][V];      // r_dif            -> p = r_dif*denom
    F al[K][V];      // t_dif            -> alpha = t_dif*denom
    F sb[K][V];      // source_up        -> beta -> src at level t0+j
    F qb[K][V];      // source_dn        -> q = source_dn*denom -> b

    // ---- (a) two-stream coefficients: every layer independent of the others (the direct-beam chain comes afterwards, so
    //      that the compiler need not keep r_dir, t_dir, t_noscat of all K layers alive next to the result arrays)
    #pragma unroll
    for (int j=0; j<K; ++j)
    {
        // keep at most LOADG layers of loads in flight per wave: the whole burst (3*K loads) would pin 6*K VGPRs
        if (j % LOADG == 0) __builtin_amdgcn_sched_barrier(0);
        const int s = t0 + j;
        const bool valid = s < nlay;
        const int sc = min(s, nlay-1);
        const int ml = top_at_1 ? sc : nlay-1-sc;
        size_t off = lay_base + size_t(ml)*ncl;
        Vec<F,V> tv, wv, gv;
        if constexpr (PRE)
        {
            tv = nt[j]; wv = nw[j];
            if constexpr (GZ)
            {
                #pragma unroll
                for (int v=0; v<V; ++v) gv.v[v] = F(0.);
            }
            else gv = ng[j];
        }
        else
        {
        // Fused broadband form only (register budget: the g-point sums live across the whole body): empty asm statements
        // tie this layer's loads to the result of layer j-BB_LOADS and each evaluation to the result BB_EVALS evaluations back,
        // so that at most that many layers of loads / two_stream temporaries are live at once. The per-g-point form
        // leaves the scheduler free (measured: 6.9 ms free vs 8.5 ms chained at 16384x140x224; fused 6.7 vs 14.7 ms).
        if constexpr (BB)
        {
            if (j >= BB_LOADS) asm volatile("" : "+v"(off) : "v"(qb[j-BB_LOADS][V-1]));
        }
        tv = load_cols<F,V>(tau + off);
        wv = load_cols<F,V>(ssa + off);
        if constexpr (GZ)                                  // asymmetry identically zero (clear-sky gas optics): g is not read
        {
            #pragma unroll
            for (int v=0; v<V; ++v) gv.v[v] = F(0.);
        }
        else gv = load_cols<F,V>(g + off);
        }
        // (V == 1: the tie sits ahead of the evaluation loop, V > 1: on each column's tau. Same dependence, but the
        //  register allocator lands differently: measured fp64 6.3 vs 8.3 ms and fp32 6.4 vs 4.4 ms, tools/ab_sw.sh)
        if constexpr (BB && V == 1)
        {
            if (j >= BB_EVALS) asm volatile("" : "+v"(tv.v[0]) : "v"(qb[j-BB_EVALS][0]));
        }
        #pragma unroll
        for (int v=0; v<V; ++v)
        {
            if constexpr (BB && V > 1)
            {
                const int e = j*V + v - BB_EVALS;             // the evaluation this one waits for
                if (e >= 0) asm volatile("" : "+v"(tv.v[v]) : "v"(qb[e / V][e % V]));
            }
            // a padding layer (level slot beyond the surface) is made transparent through its optical depth: tau = 0 gives
            // r_dif = 0, t_noscat = 1 exactly, t_dif = 1 to an ulp and r_dir = t_dir = the eps floor of the clamps (2e-16 of the
            // direct beam) -- one select on the input instead of five on the outputs
            const TwoStream<F> ts = two_stream<F,GZ>(valid ? tv.v[v] : F(0.), wv.v[v], gv.v[v], mu.v[v], mu_inv[v]);
            rp[j][v] = ts.r_dif;
            al[j][v] = ts.t_dif;
            sb[j][v] = ts.r_dir;
            qb[j][v] = ts.t_dir;
            lds_dir[j*V+v][tid] = ts.t_noscat;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    RRX_SW_T(0)

    // ---- (b) direct beam relative to the lane's incoming beam: prefix products of t_noscat, kept in LDS
    F Tloc[V];
    #pragma unroll
    for (int v=0; v<V; ++v)
    {
        F T = F(1.);
        #pragma unroll
        for (int j=0; j<K; ++j)
        {
            const F tn = lds_dir[j*V+v][tid];
            lds_dir[j*V+v][tid] = T;
            sb[j][v] *= T;
            qb[j][v] *= T;
            T *= tn;
        }
        Tloc[v] = T;
    }

    Vec<F,V> inc_dir, a_dir, a_dif, inc_dif;
    if constexpr (PRE)
    {
        inc_dir = n_inc; a_dir = n_adir; a_dif = n_adif;
        if (inc_flux_dif != nullptr) inc_dif = load_cols<F,V>(inc_flux_dif + sfc_idx);     // rare: not worth registers across the loop
    }
    else
    {
        inc_dir = load_cols<F,V>(inc_flux_dir + sfc_idx);
        a_dir = load_cols<F,V>(sfc_alb_dir + sfc_idx);
        a_dif = load_cols<F,V>(sfc_alb_dif + sfc_idx);
        if (inc_flux_dif != nullptr) inc_dif = load_cols<F,V>(inc_flux_dif + sfc_idx);
    }

    F dn_in[V], dir_in[V];

    #pragma unroll
    for (int v=0; v<V; ++v)
    {
        // ---- direct beam: inclusive product scan over level-lanes 0..ll
        F pr = Tloc[v];
        #pragma unroll
        for (int d=1; d<LL; d<<=1)
        {
            const F p2 = shfl(pr, lane - d*CL);
            if (ll >= d) pr *= p2;
        }
        F pe = shfl(pr, lane - CL);
        if (ll == 0) pe = F(1.);
        F ptot = shfl(pr, (LL-1)*CL + cl);
        if constexpr (W >= 2)
        {
            if (ll == LL-1) xch[8*v+0][wave][cl] = pr;
            RRX_SW_T(1)
            __syncthreads();
            RRX_SW_T(6)
            if constexpr (PRE)
            {
                if (v == 0)
                {
                    // every wave of the workgroup is here: the two waves that share each 128-B line ask for it together
                    __builtin_amdgcn_sched_barrier(0);
                    const int gn = min(igpt + 1, g_end - 1);          // (last iteration: a harmless re-read)
                    const F* __restrict__ tau_n = tau + size_t(gn)*ncl*nlay;
                    const F* __restrict__ ssa_n = ssa + size_t(gn)*ncl*nlay;
                    #pragma unroll
                    for (int j=0; j<K; ++j) { const unsigned o = off_of(j); nt[j] = load_cols<F,V>(tau_n + o); nw[j] = load_cols<F,V>(ssa_n + o); }
                    if constexpr (!GZ)
                    {
                        const F* __restrict__ g_n = g + size_t(gn)*ncl*nlay;
                        #pragma unroll
                        for (int j=0; j<K; ++j) ng[j] = load_cols<F,V>(g_n + off_of(j));
                    }
                    const size_t sn = size_t(gn)*ncl + icol;
                    n_inc = load_cols<F,V>(inc_flux_dir + sn); n_adir = load_cols<F,V>(sfc_alb_dir + sn); n_adif = load_cols<F,V>(sfc_alb_dif + sn);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (W == 2)
            {
                const F other = xch[8*v+0][wave^1][cl];
                if (h == 1) pe *= other;
                ptot *= other;
            }
            else
            {
                F above = F(1.), all = F(1.);
                #pragma unroll
                for (int w=0; w<W; ++w)
                {
                    const F o = xch[8*v+0][w0+w][cl];
                    if (w < h) above *= o;
                    all *= o;
                }
                pe *= above;
                ptot = all;
            }
        }
        const F dir_top = inc_dir.v[v] * mu.v[v];
        dir_in[v] = dir_top * pe;
        const F dir_sfc = dir_top * ptot;
        #pragma unroll
        for (int j=0; j<K; ++j)
        {
            sb[j][v] *= dir_in[v];
            qb[j][v] *= dir_in[v];
        }

        // ---- albedo: Moebius composite of this lane's layers (layer K-1 applied first), normalised to m11 = 1
        F m00 = F(1.), m01 = F(0.), m10 = F(0.), m11 = F(1.);
        #pragma unroll
        for (int j=K-1; j>=0; --j)
        {
            const F r = rp[j][v], t = al[j][v];
            const F e = t*t - r*r;
            const F n00 = e*m00 + r*m10, n01 = e*m01 + r*m11;
            const F n10 = m10 - r*m00,   n11 = m11 - r*m01;
            m00 = n00; m01 = n01; m10 = n10; m11 = n11;
        }
        {
            const F inv = fast_rcp(m11);
            m00 *= inv; m01 *= inv; m10 *= inv; m11 = F(1.);
        }
        // inclusive suffix scan: S(ll) = M_ll * M_{ll+1} * ... * M_7
        #pragma unroll
        for (int d=1; d<LL; d<<=1)
        {
            const F p00 = shfl(m00, lane + d*CL), p01 = shfl(m01, lane + d*CL);
            const F p10 = shfl(m10, lane + d*CL);            // partner m11 == 1
            if (ll + d < LL)
            {
                const F n00 = m00*p00 + m01*p10, n01 = m00*p01 + m01;
                const F n10 = m10*p00 + p10,     n11 = m10*p01 + F(1.);
                const F inv = fast_rcp(n11);
                m00 = n00*inv; m01 = n01*inv; m10 = n10*inv;
            }
        }
        F x00 = F(1.), x01 = F(0.), x10 = F(0.);      // composite of everything below this wave's levels
        RRX_SW_T(2)
        if constexpr (W == 2)
        {
            if (ll == 0) { xch[8*v+1][wave][cl] = m00; xch[8*v+2][wave][cl] = m01; xch[8*v+3][wave][cl] = m10; }
            __syncthreads();
            RRX_SW_T(6)
            if (h == 0)
            {
                x00 = xch[8*v+1][wave^1][cl]; x01 = xch[8*v+2][wave^1][cl]; x10 = xch[8*v+3][wave^1][cl];
                const F n00 = m00*x00 + m01*x10, n01 = m00*x01 + m01;
                const F n10 = m10*x00 + x10,     n11 = m10*x01 + F(1.);
                const F inv = fast_rcp(n11);
                m00 = n00*inv; m01 = n01*inv; m10 = n10*inv;
            }
        }
        else if constexpr (W > 2)
        {
            if (ll == 0) { xch[8*v+1][wave][cl] = m00; xch[8*v+2][wave][cl] = m01; xch[8*v+3][wave][cl] = m10; }
            __syncthreads();
            RRX_SW_T(6)
            // composite of the waves below this one (the lowest applied first), then this wave's on top of it
            #pragma unroll
            for (int w=W-1; w>=1; --w)
                if (w > h)
                {
                    const F o00 = xch[8*v+1][w0+w][cl], o01 = xch[8*v+2][w0+w][cl], o10 = xch[8*v+3][w0+w][cl];
                    const F n00 = o00*x00 + o01*x10, n01 = o00*x01 + o01;
                    const F n10 = o10*x00 + x10,     n11 = o10*x01 + F(1.);
                    const F inv = fast_rcp(n11);
                    x00 = n00*inv; x01 = n01*inv; x10 = n10*inv;
                }
            if (h < W-1)
            {
                const F n00 = m00*x00 + m01*x10, n01 = m00*x01 + m01;
                const F n10 = m10*x00 + x10,     n11 = m10*x01 + F(1.);
                const F inv = fast_rcp(n11);
                m00 = n00*inv; m01 = n01*inv; m10 = n10*inv;
            }
        }
        F e00 = shfl(m00, lane + CL), e01 = shfl(m01, lane + CL), e10 = shfl(m10, lane + CL);
        if (ll == LL-1) { e00 = x00; e01 = x01; e10 = x10; }
        const F alb_sfc = a_dif.v[v];
        F a = (e00*alb_sfc + e01) * fast_rcp(e10*alb_sfc + F(1.));      // albedo at the bottom of this lane's chunk

        // ---- replay albedo upward; build alpha, beta, p, q and the lane's affine composites
        F As = F(1.), Bs = F(0.), Bd = F(0.);    // As: product of alpha (shared); Bs: source (upward); Bd: down
        #pragma unroll
        for (int j=K-1; j>=0; --j)
        {
            const F r = rp[j][v], t = al[j][v];
            const F denom = fast_rcp(F(1.) - r*a);
            const F alpha = t*denom;
            const F beta = sb[j][v] + alpha*a*qb[j][v];
            a = r + t*alpha*a;
            lds_alb[j*V+v][tid] = a;
            al[j][v] = alpha;
            sb[j][v] = beta;
            rp[j][v] = r*denom;
            qb[j][v] = qb[j][v]*denom;
            Bs = alpha*Bs + beta;
            As *= alpha;
        }

        // ---- source: suffix affine scan (lanes below applied first)
        F sa = As, sbb = Bs;
        #pragma unroll
        for (int d=1; d<LL; d<<=1)
        {
            const F a2 = shfl(sa, lane + d*CL), b2 = shfl(sbb, lane + d*CL);
            if (ll + d < LL) { sbb = sa*b2 + sbb; sa = sa*a2; }
        }
        F xa = F(1.), xb = F(0.);
        RRX_SW_T(3)
        if constexpr (W == 2)
        {
            if (ll == 0) { xch[8*v+4][wave][cl] = sa; xch[8*v+5][wave][cl] = sbb; }
            __syncthreads();
            RRX_SW_T(6)
            if (h == 0)
            {
                xa = xch[8*v+4][wave^1][cl]; xb = xch[8*v+5][wave^1][cl];
                sbb = sa*xb + sbb; sa = sa*xa;
            }
        }
        else if constexpr (W > 2)
        {
            if (ll == 0) { xch[8*v+4][wave][cl] = sa; xch[8*v+5][wave][cl] = sbb; }
            __syncthreads();
            RRX_SW_T(6)
            #pragma unroll
            for (int w=W-1; w>=1; --w)
                if (w > h) { const F oa = xch[8*v+4][w0+w][cl], ob = xch[8*v+5][w0+w][cl]; xb = oa*xb + ob; xa = oa*xa; }
            if (h < W-1) { sbb = sa*xb + sbb; sa = sa*xa; }
        }
        F ae = shfl(sa, lane + CL), be = shfl(sbb, lane + CL);
        if (ll == LL-1) { ae = xa; be = xb; }
        const F src_sfc = dir_sfc * a_dir.v[v];
        F s = ae*src_sfc + be;                                   // src at the bottom of this lane's chunk

        // replay src upward; b_j = p_j*src_below + q_j; accumulate the downward composite
        F Q = F(1.);
        #pragma unroll
        for (int j=K-1; j>=0; --j)
        {
            const F b = rp[j][v]*s + qb[j][v];
            s = al[j][v]*s + sb[j][v];
            sb[j][v] = s;
            qb[j][v] = b;
            Bd += Q*b;
            Q *= al[j][v];
        }

        // ---- diffuse down: prefix affine scan
        F da = As, db = Bd;
        #pragma unroll
        for (int d=1; d<LL; d<<=1)
        {
            const F a2 = shfl(da, lane - d*CL), b2 = shfl(db, lane - d*CL);
            if (ll >= d) { db = da*b2 + db; da = da*a2; }
        }
        xa = F(1.); xb = F(0.);
        RRX_SW_T(4)
        if constexpr (W == 2)
        {
            if (ll == LL-1) { xch[8*v+6][wave][cl] = da; xch[8*v+7][wave][cl] = db; }
            __syncthreads();
            RRX_SW_T(6)
            if (h == 1)
            {
                xa = xch[8*v+6][wave^1][cl]; xb = xch[8*v+7][wave^1][cl];
                db = da*xb + db; da = da*xa;
            }
        }
        else if constexpr (W > 2)
        {
            if (ll == LL-1) { xch[8*v+6][wave][cl] = da; xch[8*v+7][wave][cl] = db; }
            __syncthreads();
            RRX_SW_T(6)
            #pragma unroll
            for (int w=0; w<W-1; ++w)
                if (w < h) { const F oa = xch[8*v+6][w0+w][cl], ob = xch[8*v+7][w0+w][cl]; xb = oa*xb + ob; xa = oa*xa; }
            if (h > 0) { db = da*xb + db; da = da*xa; }
        }
        ae = shfl(da, lane - CL); be = shfl(db, lane - CL);
        if (ll == 0) { ae = xa; be = xb; }
        const F dn_top = (inc_flux_dif != nullptr) ? inc_dif.v[v] : F(0.);
        dn_in[v] = ae*dn_top + be;
    }

    // ---- replay the diffuse downward flux and store this lane's K levels as soon as each value exists
    F dn[V];
    #pragma unroll
    for (int v=0; v<V; ++v) dn[v] = dn_in[v];
    #pragma unroll
    for (int j=0; j<K; ++j)
    {
        Vec<F,V> ou, od, odr;
        #pragma unroll
        for (int v=0; v<V; ++v)
        {
            const F dr = dir_in[v] * lds_dir[j*V+v][tid];
            ou.v[v] = dn[v]*lds_alb[j*V+v][tid] + sb[j][v];
            od.v[v] = dn[v] + dr;
            odr.v[v] = dr;
            dn[v] = al[j][v]*dn[v] + qb[j][v];
        }
        const int t = t0 + j;
        if constexpr (BB)
        {
            #pragma unroll
            for (int v=0; v<V; ++v)
            {
                F au = lds_acc_up[j*V+v][tid], ad = lds_acc_dn[j*V+v][tid];
                add_rounded(au, ou.v[v]); add_rounded(ad, od.v[v]); add_rounded(acc_dir[j][v], odr.v[v]);
                lds_acc_up[j*V+v][tid] = au; lds_acc_dn[j*V+v][tid] = ad;
            }
        }
        else if (writer && t <= nlay)
        {
            const int ml = top_at_1 ? t : nlay - t;
            const size_t o = lev_base + size_t(ml)*ncl;
            store_cols<F,V>(flux_up + o, ou);
            store_cols<F,V>(flux_dn + o, od);
            store_cols<F,V>(flux_dir + o, odr);
        }
    }
    RRX_SW_T(5)
    }   // g-point loop
#if RRX_SW_TIMING
    if (lane == 0) for (int k=0; k<8; ++k) atomicAdd(&g_sw_clk[wave & 15][k], sw_acc[k]);
#endif

    if constexpr (BB)
    {
        if (!writer) return;
        #pragma unroll
        for (int j=0; j<K; ++j)
        {
            const int t = t0 + j;
            if (t <= nlay)
            {
                const int ml = top_at_1 ? t : nlay - t;
                const size_t o = size_t(icol) + size_t(ml)*ncl + (GS ? size_t(blockIdx.y)*ncl*nlev : size_t(0));
                Vec<F,V> u, d, r;
                #pragma unroll
                for (int v=0; v<V; ++v) { u.v[v] = lds_acc_up[j*V+v][tid]; d.v[v] = lds_acc_dn[j*V+v][tid]; r.v[v] = acc_dir[j][v]; }
                store_cols<F,V>(flux_up + o, u);
                store_cols<F,V>(flux_dn + o, d);
                store_cols<F,V>(flux_dir + o, r);
            }
        }
    }
}


// Any-nlay fallback: one thread per (col, gpt); r_dif, t_dif, source_up/dn, albedo, src, denom kept in a
// caller-provided global workspace laid out like the reference's temporaries.
template<typename F>
__global__ void __launch_bounds__(256)
sw_2stream_serial_kernel(
        const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* __restrict__ tau, const F* __restrict__ ssa, const F* __restrict__ g, const F* __restrict__ mu0,
        const F* __restrict__ sfc_alb_dir, const F* __restrict__ sfc_alb_dif,
        const F* __restrict__ inc_flux_dir, const F* __restrict__ inc_flux_dif,
        F* __restrict__ flux_up, F* __restrict__ flux_dn, F* __restrict__ flux_dir,
        F* __restrict__ ws)
{
    const int icol = blockIdx.x*blockDim.x + threadIdx.x;
    const int igpt = blockIdx.y;
    if (icol >= ncol) return;

    const int nlev = nlay+1;
    const size_t ncl = size_t(ncol);
    const size_t lay_base = size_t(igpt)*ncl*nlay + icol;
    const size_t lev_base = size_t(igpt)*ncl*nlev + icol;
    const size_t sfc_idx = size_t(igpt)*ncl + icol;
    const size_t opt = ncl*nlay*ngpt;
    F* w_r = ws; F* w_t = ws + opt; F* w_su = ws + 2*opt; F* w_sd = ws + 3*opt; F* w_den = ws + 4*opt;
    F* w_alb = ws + 5*opt; F* w_src = w_alb + ncl*nlev*ngpt;

    auto mlev = [&](const int t) { return lev_base + size_t(top_at_1 ? t : nlay - t)*ncl; };
    auto mlay = [&](const int s) { return lay_base + size_t(top_at_1 ? s : nlay-1-s)*ncl; };

    const F mu = mu0[icol];
    F dir = inc_flux_dir[sfc_idx] * mu;
    for (int s=0; s<nlay; ++s)
    {
        const size_t il = mlay(s);
        const TwoStream<F> ts = two_stream<F>(tau[il], ssa[il], g[il], mu, F(1.)/mu);
        w_r[il] = ts.r_dif; w_t[il] = ts.t_dif;
        w_su[il] = ts.r_dir * dir; w_sd[il] = ts.t_dir * dir;
        flux_dir[mlev(s)] = dir;
        dir *= ts.t_noscat;
    }
    flux_dir[mlev(nlay)] = dir;

    F a = sfc_alb_dif[sfc_idx];
    F sr = dir * sfc_alb_dir[sfc_idx];
    w_alb[mlev(nlay)] = a; w_src[mlev(nlay)] = sr;
    for (int s=nlay-1; s>=0; --s)
    {
        const size_t il = mlay(s);
        const F r = w_r[il], t = w_t[il];
        const F denom = F(1.)/(F(1.) - r*a);
        w_den[il] = denom;
        sr = w_su[il] + t*denom*(sr + a*w_sd[il]);
        a = r + t*t*a*denom;
        w_alb[mlev(s)] = a; w_src[mlev(s)] = sr;
    }

    F dn = (inc_flux_dif != nullptr) ? inc_flux_dif[sfc_idx] : F(0.);
    flux_up[mlev(0)] = dn*a + sr;
    flux_dn[mlev(0)] = dn + flux_dir[mlev(0)];
    for (int s=0; s<nlay; ++s)
    {
        const size_t il = mlay(s);
        const size_t lv = mlev(s+1);
        dn = (w_t[il]*dn + w_r[il]*w_src[lv] + w_sd[il]) * w_den[il];
        flux_up[lv] = dn*w_alb[lv] + w_src[lv];
        flux_dn[lv] = dn + flux_dir[lv];
    }
}


template<typename F>
__global__ void sum_gpt_kernel(const size_t ncl_lev, const int ngpt, const F* __restrict__ in, F* __restrict__ out)
{
    const size_t i = size_t(blockIdx.x)*blockDim.x + threadIdx.x;
    if (i >= ncl_lev) return;
    F s = F(0.);
    for (int ig=0; ig<ngpt; ++ig) s += in[i + size_t(ig)*ncl_lev];
    out[i] = s;
}

// the partial sums of the g-point ranges of a fused broadband launch, all flux arrays in one launch (blockIdx.y = array; the
// partials of array a start at in + a*nsplit*ncl_lev): range order, as sum_gpt_kernel
template<typename F, int NARR>
__global__ void sum_ranges_kernel(const size_t ncl_lev, const int nsplit, const F* __restrict__ in, F* const o0, F* const o1, F* const o2)
{
    const size_t i = size_t(blockIdx.x)*blockDim.x + threadIdx.x;
    if (i >= ncl_lev) return;
    const int a = blockIdx.y;
    const F* __restrict__ p = in + size_t(a)*nsplit*ncl_lev;
    F s = F(0.);
    for (int ig=0; ig<nsplit; ++ig) s += p[i + size_t(ig)*ncl_lev];
    F* __restrict__ out = (a == 0) ? o0 : ((a == 1 || NARR < 3) ? o1 : o2);
    out[i] = s;
}

template<typename F>
__global__ void apply_BC_kernel(const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* __restrict__ inc_flux, const F* __restrict__ factor, F* __restrict__ flux_dn)
{
    const int icol = blockIdx.x*blockDim.x + threadIdx.x;
    const int igpt = blockIdx.y;
    if (icol >= ncol) return;
    const size_t o = size_t(icol) + size_t(top_at_1 ? 0 : nlay)*ncol + size_t(igpt)*ncol*(nlay+1);
    F v = F(0.);
    if (inc_flux != nullptr) v = inc_flux[icol + size_t(igpt)*ncol];
    if (factor != nullptr) v *= factor[icol];
    flux_dn[o] = v;
}

template<typename F, int V, int W>
bool launch_scan(hipStream_t st,
        const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* tau, const F* ssa, const F* g, const F* mu0, const F* sfc_alb_dir, const F* sfc_alb_dif,
        const F* inc_flux_dir, const F* inc_flux_dif, F* flux_up, F* flux_dn, F* flux_dir)
{
    const dim3 grid(ceil_div(ncol, (4/W)*CL*V), ngpt);
    const int need = ceil_div(nlay+1, LL*W);
#define RRX_SW_K(KK) if (need <= KK) { sw_2stream_scan_kernel<F,V,KK,W><<<grid, 256, 0, st>>>( \
        ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif, inc_flux_dir, inc_flux_dif, \
        flux_up, flux_dn, flux_dir, tuning().sync_waves, 1); return true; }
    if constexpr (W == 1) { RRX_SW_K(4) RRX_SW_K(8) RRX_SW_K(12) RRX_SW_K(18) RRX_SW_K(24) RRX_SW_K(33) }
    else                  { RRX_SW_K(2) RRX_SW_K(4) RRX_SW_K(6)  RRX_SW_K(9)  RRX_SW_K(12) RRX_SW_K(17) }
#undef RRX_SW_K
    return false;
}

// run f with a compile-time copy of a run-time flag
template<typename Fn> void with_flag(const bool flag, Fn&& f) { if (flag) f(std::true_type{}); else f(std::false_type{}); }

// Fused broadband form. W waves per column group, CLT column lanes per wave (two column groups per workgroup); false when the
// columns are taller than the form's largest K (the caller tries the next form).
template<typename F, int V, int W = 2, int CLT = 8, int NWG = (W > 2 ? 2*W : 4)>
bool launch_scan_bb(hipStream_t st,
        const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* tau, const F* ssa, const F* g, const F* mu0, const F* sfc_alb_dir, const F* sfc_alb_dif,
        const F* inc_flux_dir, const F* inc_flux_dif, F* flux_up, F* flux_dn, F* flux_dir)
{
    constexpr int NW = NWG;                               // wavefronts per workgroup: two column groups (one where NWG == W)
    constexpr int KMAX = (CLT == 16) ? 12 : (W > 2 ? 9 : 12);       // (8 x 8 lanes, W = 4: nine layers per lane fill the LDS of a CU)
    const int groups = ceil_div(ncol, (NW/W)*CLT*V);
    const int need = ceil_div(nlay+1, (64/CLT)*W);
    if (need > KMAX) return false;
    // pipelined loads: fp64 without g array only (with it 27 prefetched doubles spill); fp32 in the one-column-per-lane geometry
    // (two columns per lane: 140 B of scratch per lane, 4.00 against 3.63 ms at C4)
    const bool pre = tuning().sw_variant != 8 && size_t(ncol)*nlay < (size_t(1) << 31)
                     && (sizeof(F) == 8 ? g == nullptr : (V == 1 && CLT == 16));
    // few column groups: the g-point loop is split over grid.y, partial sums added in range order afterwards
    const int gper = ceil_div(ngpt, broadband_gsplit(groups, ngpt, (NW > 4) ? 256 : ((CLT == 16) ? 256*RRX_SW_F32_WAVES1 : 512)));      // (one or two workgroups per CU)
    const int nsplit = ceil_div(ngpt, gper);               // no empty range: every workgroup's first g-point exists (it is prefetched)
    const size_t nlevcol = size_t(ncol)*(nlay+1);
    StreamScratch scratch(st);
    F* up = flux_up; F* dn = flux_dn; F* dr = flux_dir;
    if (nsplit > 1) { up = scratch.get<F>(3*nsplit*nlevcol); dn = up + nsplit*nlevcol; dr = dn + nsplit*nlevcol; }
    const dim3 grid(groups, nsplit);
    const int sync_waves = tuning().sync_waves;
    auto launch = [&](auto kk)
    {
        constexpr int KK = decltype(kk)::value;
        with_flag(g == nullptr, [&](auto gz) { with_flag(pre, [&](auto pr) { with_flag(nsplit > 1, [&](auto gs)
        {
            constexpr bool GZ = decltype(gz)::value, GS = decltype(gs)::value;
            // (the pipelined form exists where launch_scan_bb may pick it: fp64 without g array, fp32 with one column per lane)
            constexpr bool PRE = decltype(pr)::value && (sizeof(F) == 8 ? GZ : (V == 1 && CLT == 16));
            sw_2stream_scan_kernel<F,V,KK,W,true,GZ,PRE,GS,NW,CLT><<<grid, 64*NW, 0, st>>>(
                ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif, inc_flux_dir, inc_flux_dif,
                up, dn, dr, sync_waves, gper);
        }); }); });
    };
#define RRX_SW_K(KK) if (need <= KK) { launch(std::integral_constant<int,KK>{}); break; }
    do
    {
        if constexpr (CLT == 16) { RRX_SW_K(2) RRX_SW_K(4) RRX_SW_K(6) RRX_SW_K(9) RRX_SW_K(12) }
        else if constexpr (W == 2) { RRX_SW_K(2) RRX_SW_K(4) RRX_SW_K(6) RRX_SW_K(9) RRX_SW_K(12) }
        else if constexpr (W == 8) { RRX_SW_K(5) RRX_SW_K(7) RRX_SW_K(9) }      // (288 ... 319 / 447 / 575 layers)
        else { RRX_SW_K(9) }
    } while (false);
#undef RRX_SW_K
    if (nsplit > 1)      // (up, dn, dr lie behind each other in the scratch block)
        sum_ranges_kernel<F,3><<<dim3(ceil_div(nlevcol, 256), 3), 256, 0, st>>>(nlevcol, nsplit, up, flux_up, flux_dn, flux_dir);
    return true;
}

template<typename F>
int sw_solver_2stream_impl(
        const int ncol, const int nlay, const int ngpt, const Bool top_at_1,
        const F* tau, const F* ssa, const F* g, const F* mu0,
        const F* sfc_alb_dir, const F* sfc_alb_dif, const F* inc_flux_dir,
        F* flux_up, F* flux_dn, F* flux_dir,
        const Bool has_dif_bc, const F* inc_flux_dif,
        const Bool do_broadband, F* flux_up_loc, F* flux_dn_loc, F* flux_dir_loc, void* stream)
{
    RRX_TRY
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ncol <= 0 || nlay <= 0 || ngpt <= 0) throw std::runtime_error("empty problem");
    const F* dif = has_dif_bc ? inc_flux_dif : nullptr;
    const int g_sw_variant = tuning().sw_variant;
    const int g_bb_min_groups = tuning().bb_min_groups;

    // broadband mode, fused form (see the kernel's BB note): one workgroup per column group sums all g-points in order when the
    // column groups alone fill the chip, otherwise the g-point range is split over grid.y (rrx::broadband_gsplit)
    constexpr int VBB = (sizeof(F) == 8) ? 1 : 2;
    (void)g_bb_min_groups;
    // (variant 8: fused broadband form without the pipelined loads, for A/B runs)
    if (do_broadband && g_sw_variant != 1 && g_sw_variant != 7 && (ncol % VBB == 0 || sizeof(F) == 4))
    {
        if (flux_up_loc == nullptr || flux_dn_loc == nullptr || flux_dir_loc == nullptr)
            throw std::runtime_error("do_broadband needs flux_*_loc");
        // fp32: one column per lane, 16 x 4 lanes, four waves per column group (up to 191 layers); variant 9 = the two-columns-
        // per-lane form of rounds 1-3 for A/B runs
        if constexpr (sizeof(F) == 4)
        {
            // up to 143 layers (nine per lane): one column group per workgroup at three waves per SIMD; 144-191 (twelve per lane, which
            // spills at 168 VGPRs): two groups per workgroup at two waves per SIMD
            if (g_sw_variant != 9 && ceil_div(nlay+1, 16) <= 9 &&
                launch_scan_bb<F,1,4,16,RRX_SW_F32_NW>(st, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif,
                                                       inc_flux_dir, dif, flux_up_loc, flux_dn_loc, flux_dir_loc))
                return 0;
            if (g_sw_variant != 9 &&
                launch_scan_bb<F,1,4,16>(st, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif,
                                         inc_flux_dir, dif, flux_up_loc, flux_dn_loc, flux_dir_loc))
                return 0;
        }
        if (ncol % VBB == 0 &&
            launch_scan_bb<F,VBB>(st, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif,
                                  inc_flux_dir, dif, flux_up_loc, flux_dn_loc, flux_dir_loc))
            return 0;
        // 192 ... 287 layers: four wavefronts per column group
        if (ncol % VBB == 0 &&
            launch_scan_bb<F,VBB,4>(st, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif,
                                    inc_flux_dir, dif, flux_up_loc, flux_dn_loc, flux_dir_loc))
            return 0;
        // 288 ... 575 layers (round 4): eight wavefronts on ONE column group per workgroup (64 levels per wave at nine layers per lane;
        // the row segments are 64 B with no partner group in the workgroup: twice the L2 fetches, on a kernel bound by fp64 issue)
        if (ncol % VBB == 0 &&
            launch_scan_bb<F,VBB,8,8,8>(st, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif,
                                        inc_flux_dir, dif, flux_up_loc, flux_dn_loc, flux_dir_loc))
            return 0;
    }

    // g == nullptr (asymmetry identically zero) is native to the fused broadband kernels only: the other forms read zeros
    // the large temporaries of these forms come from one cached block per stream (rrx::cached_workspace): [g zeros][3 per-g-point
    // flux arrays][the serial kernel's seven cell / level arrays]
    const size_t nlevcol = size_t(ncol)*(nlay+1);
    const size_t w_g = (g == nullptr) ? size_t(ncol)*nlay*ngpt : 0, w_flux = do_broadband ? 3*nlevcol*ngpt : 0;
    const size_t w_serial = 5*size_t(ncol)*nlay*ngpt + 2*nlevcol*ngpt;
    // (one lease per call: the serial kernel's part is asked for up front whenever it could be needed)
    WorkspaceLease lease(st);
    const bool may_go_serial = g_sw_variant == 1 || (g_sw_variant != 2 ? ceil_div(nlay+1, LL*2) > 17 : ceil_div(nlay+1, LL) > 33);   // (largest K of launch_scan)
    F* big = (w_g + w_flux > 0 || may_go_serial) ? lease.get<F>(w_g + w_flux + (may_go_serial ? w_serial : 0)) : nullptr;
    if (g == nullptr)
    {
        if (hipMemsetAsync(big, 0, w_g*sizeof(F), st) != hipSuccess) throw std::runtime_error("workspace memset failed");
        g = big;
    }

    F* up = flux_up; F* dn = flux_dn; F* dr = flux_dir;
    if (do_broadband)
    {
        if (flux_up_loc == nullptr || flux_dn_loc == nullptr || flux_dir_loc == nullptr)
            throw std::runtime_error("do_broadband needs flux_*_loc");
        F* ws = big + w_g;
        up = ws; dn = ws + nlevcol*ngpt; dr = ws + 2*nlevcol*ngpt;
    }

    // g_sw_variant: 0 = default (two-wave level split), 1 = serial fallback, 2 = one wave per column group, 3 = two waves
    bool done = false;
    if (g_sw_variant != 1)
    {
#define RRX_SW_ARGS st, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif, inc_flux_dir, dif, up, dn, dr
        const bool two = (g_sw_variant != 2);
        if constexpr (sizeof(F) == 4)
        {
            if (g_sw_variant == 4 && ncol % 4 == 0) done = launch_scan<F,4,2>(RRX_SW_ARGS);
            else if (ncol % 2 == 0)
                done = two ? launch_scan<F,2,2>(RRX_SW_ARGS) : launch_scan<F,2,1>(RRX_SW_ARGS);
        }
        if (!done)
            done = two ? launch_scan<F,1,2>(RRX_SW_ARGS) : launch_scan<F,1,1>(RRX_SW_ARGS);
#undef RRX_SW_ARGS
    }
    if (!done)
    {
        if (!may_go_serial) throw std::runtime_error("internal: no tiling for this shape and no workspace for the serial form");
        F* ws2 = big + w_g + w_flux;
        const dim3 grid(ceil_div(ncol, 256), ngpt);
        sw_2stream_serial_kernel<F><<<grid, 256, 0, st>>>(ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0,
                sfc_alb_dir, sfc_alb_dif, inc_flux_dir, dif, up, dn, dr, ws2);
    }

    if (do_broadband)
    {
        const int nb = ceil_div(nlevcol, 256);
        sum_gpt_kernel<F><<<nb, 256, 0, st>>>(nlevcol, ngpt, up, flux_up_loc);
        sum_gpt_kernel<F><<<nb, 256, 0, st>>>(nlevcol, ngpt, dn, flux_dn_loc);
        sum_gpt_kernel<F><<<nb, 256, 0, st>>>(nlevcol, ngpt, dr, flux_dir_loc);
    }
    RRX_CATCH("rrx_sw_solver_2stream")
}

template<typename F>
int apply_BC_impl(int ncol, int nlay, int ngpt, Bool top_at_1, const F* inc, const F* factor, F* flux, void* stream)
{
    RRX_TRY
    apply_BC_kernel<F><<<dim3(ceil_div(ncol, 256), ngpt), 256, 0, static_cast<hipStream_t>(stream)>>>(
            ncol, nlay, ngpt, top_at_1, inc, factor, flux);
    RRX_CATCH("rrx_apply_BC")
}
}  // namespace


extern "C"
{
int rrx_set_sw_variant(int v) { rrx::tuning().sw_variant = v; return 0; }
#if RRX_SW_TIMING
// diagnostic build only: phase clocks per wavefront of a workgroup (out[16][8]) summed over the workgroups since the last call (two-stream, direct beam, albedo, source, down scan,
// final replay, barrier waits, loop top), then reset
int rrx_sw_timing(unsigned long long* out)
{
    unsigned long long zero[16*8] = {0};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sw_clk), 16*8*sizeof(unsigned long long)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sw_clk), zero, sizeof(zero)) == hipSuccess ? 0 : 1;
}
#endif
int rrx_set_broadband_min_groups(int n) { rrx::tuning().bb_min_groups = n; return 0; }
int rrx_set_broadband_gsplit(int n) { rrx::tuning().bb_gsplit = n; return 0; }

#define RRX_DEFINE_SW(F, SFX) \
int rrx_sw_solver_2stream##SFX( \
        int ncol, int nlay, int ngpt, RrxBool top_at_1, \
        const F* tau, const F* ssa, const F* g, const F* mu0, \
        const F* sfc_alb_dir, const F* sfc_alb_dif, const F* inc_flux_dir, \
        F* flux_up, F* flux_dn, F* flux_dir, \
        RrxBool has_dif_bc, const F* inc_flux_dif, \
        RrxBool do_broadband, F* flux_up_loc, F* flux_dn_loc, F* flux_dir_loc, void* stream) \
{ \
    return sw_solver_2stream_impl<F>(ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif, inc_flux_dir, \
            flux_up, flux_dn, flux_dir, has_dif_bc, inc_flux_dif, do_broadband, flux_up_loc, flux_dn_loc, flux_dir_loc, stream); \
} \
int rrx_apply_BC_factor##SFX(int ncol, int nlay, int ngpt, RrxBool top_at_1, const F* inc_flux_dir, const F* mu0, F* gpt_flux_dir, void* stream) \
{ return apply_BC_impl<F>(ncol, nlay, ngpt, top_at_1, inc_flux_dir, mu0, gpt_flux_dir, stream); } \
int rrx_apply_BC_0##SFX(int ncol, int nlay, int ngpt, RrxBool top_at_1, F* gpt_flux_dn, void* stream) \
{ return apply_BC_impl<F>(ncol, nlay, ngpt, top_at_1, (const F*)nullptr, (const F*)nullptr, gpt_flux_dn, stream); } \
int rrx_apply_BC_gpt##SFX(int ncol, int nlay, int ngpt, RrxBool top_at_1, const F* inc_flux_dif, F* gpt_flux_dn, void* stream) \
{ return apply_BC_impl<F>(ncol, nlay, ngpt, top_at_1, inc_flux_dif, (const F*)nullptr, gpt_flux_dn, stream); }

RRX_DEFINE_SW(double, _f64)
RRX_DEFINE_SW(float, _f32)
}
