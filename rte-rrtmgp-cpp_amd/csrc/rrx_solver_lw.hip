// Longwave no-scattering solver, fused (boundary condition + source + both transport sweeps + quadrature
// scaling in ONE kernel, no global scratch). Replaces the reference's launcher
//   Rte_solver_kernels_cuda::lw_solver_noscat   (/root/reference/src_kernels_cuda/rte_solver_kernels_launchers.cu:61-286)
// and its kernels lw_solver_noscat_step_{1,2,3}_kernel, lw_transport_noscat_kernel, apply_BC_kernel
// (/root/reference/src_kernels_cuda/rte_solver_kernels.cu:35-193,351-387).
//
// MI355X design (DESIGN.md section "lw_solver_noscat"):
//   * arrays are (col, lay, gpt) with the column index fastest. One wavefront owns a tile of 8*V columns x all
//     levels of one g-point. The 64 lanes are 8 column-lanes x 8 level-lanes: lane = ll*8 + cl.
//   * level-lane ll owns K consecutive levels/layers (in top-to-surface "sweep" order), held in registers.
//     Every input is read exactly once from HBM, every output written exactly once.
//   * the serial vertical recurrences x' = t*x + s are affine maps; each lane composes its K maps, an 8-lane
//     Hillis-Steele scan (3 __shfl steps, stride 8 lanes) propagates the boundary values between level-lanes,
//     and each lane then replays its K layers from its incoming value.
#include "rrx_common.h"
#include "rrx_hip.h"

#pragma clang fp contract(fast)

namespace
{
using namespace rrx;

#ifndef RRX_LW_DEFAULT_VARIANT
#define RRX_LW_DEFAULT_VARIANT 5
#endif
constexpr int CL = 8;    // column lanes
constexpr int LL = 8;    // level lanes

// W = waves per column group: 1 = the whole column in one wavefront (8 level-lanes x K layers); 2 = the levels of the
// same 8*V columns spread over 16 level-lanes in two adjacent wavefronts (half the per-lane state, twice the resident
// waves); the two vertical scans then exchange each wave's total through LDS, one block barrier per scan.
// BB (broadband): the workgroup walks over ALL g-points of its columns and keeps the g-point sum of both fluxes in
// registers (same summation order as sum_broadband over stored per-g-point fluxes, so the same bits); flux_up/flux_dn
// are then (ncol, nlev) arrays. Saves the per-g-point flux stores and the reduction pass that reads them back.
// CLT = column lanes per wavefront (level lanes = 64 / CLT): 8 x 8 is the default geometry; 16 x 4 with W = 4 keeps K = 9
// layers per lane at 140 layers (the register budget of the two-wave form) and doubles the row segment of a wavefront
// (128 B in fp64 with V = 1).
template<typename F, int V, int K, int W, bool JAC, bool ACC, bool BB = false, int CLT = 8>
__global__ void __launch_bounds__(256, (W >= 2) ? 2 : 1)
lw_noscat_scan_kernel(
        const int ncol, const int nlay, const int ngpt, const int top_at_1, const int imu,
        const F* __restrict__ secants, const F* __restrict__ weights,
        const F* __restrict__ tau, const F* __restrict__ lay_source, const F* __restrict__ lev_source,
        const F* __restrict__ sfc_emis, const F* __restrict__ sfc_src, const F* __restrict__ inc_flux,
        F* __restrict__ flux_up, F* __restrict__ flux_dn,
        const F* __restrict__ sfc_src_jac, F* __restrict__ flux_up_jac, const int sync_waves)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int CL = CLT, LL = 64 / CLT;           // shadow the default geometry
    const int cl = lane & (CL-1);
    const int ll = lane / CL;
    const int h = wave % W;                          // which part of the column this wave holds (0 = TOA side)
    const int wave_col0 = (blockIdx.x*(4/W) + wave/W) * (CL*V);
    __shared__ F xch[(W >= 2) ? 4*V : 1][4][CL];     // wave totals of the two scans
    // The two waves that share each 128-B line (8 columns x 8 B = 64 B per wave when V = 1) must issue their load
    // bursts together, or the second half of every line is fetched from HBM again once L2 has turned over
    // (measured: +26 % FETCH_SIZE without the barrier). No thread leaves before the barrier.
    if constexpr (W == 1) { if (wave_col0 >= ncol) return; }   // wave-uniform; W == 2 keeps every wave for the barriers

    int icol = wave_col0 + cl*V;
    const bool active = icol < ncol;                  // all V columns exist (ncol % V == 0) or none
    if (!active) icol = (wave_col0 < ncol) ? wave_col0 : 0;    // harmless duplicate loads, no stores
    const bool writer = active && wave_col0 < ncol;

    const int nlev = nlay + 1;
    const size_t ncl = size_t(ncol);
    const int t0 = (h*LL + ll)*K;

    F acc_up[BB ? K : 1][V], acc_dn[BB ? K : 1][V];
    if constexpr (BB)
    {
        #pragma unroll
        for (int j=0; j<K; ++j)
            #pragma unroll
            for (int v=0; v<V; ++v) { acc_up[j][v] = F(0.); acc_dn[j][v] = F(0.); }
    }

    const int g_begin = BB ? 0 : blockIdx.y;
    const int g_end = BB ? ngpt : blockIdx.y + 1;
    for (int igpt=g_begin; igpt<g_end; ++igpt)
    {
    if (sync_waves) __syncthreads();
    const size_t lay_base = size_t(igpt)*ncl*nlay + icol;
    const size_t lev_base = size_t(igpt)*ncl*nlev + icol;
    const size_t sfc_idx = size_t(igpt)*ncl + icol;

    const F pi = F(3.14159265358979323846);
    const F eps = Lim<F>::eps();
    const F tau_thres = sqrt(sqrt(eps));

    const Vec<F,V> D = load_cols<F,V>(secants + sfc_idx + size_t(imu)*ncl*ngpt);
    const F w = weights[imu];

    F tr[K][V], sdn[K][V], sup[K][V];
    Vec<F,V> lv[K];

    // level sources at this lane's K levels (sweep order: level t is ABOVE layer t)
    #pragma unroll
    for (int j=0; j<K; ++j)
    {
        const int t = min(t0 + j, nlay);
        const int ml = top_at_1 ? t : nlay - t;
        lv[j] = load_cols<F,V>(lev_source + lev_base + size_t(ml)*ncl);
    }

    // level below the lane's last layer: first level of the next level-lane; across the wave seam (W == 2) it is loaded
    Vec<F,V> lv_next;
    #pragma unroll
    for (int v=0; v<V; ++v) lv_next.v[v] = shfl(lv[0].v[v], lane + CL);
    if constexpr (W >= 2)
    {
        if (ll == LL-1)
        {
            const int t = min(t0 + K, nlay);
            const int ml = top_at_1 ? t : nlay - t;
            lv_next = load_cols<F,V>(lev_source + lev_base + size_t(ml)*ncl);
        }
    }

    F A[V], Bdn[V], Bup[V];
    #pragma unroll
    for (int v=0; v<V; ++v) { A[v] = F(1.); Bdn[v] = F(0.); Bup[v] = F(0.); }

    #pragma unroll
    for (int j=0; j<K; ++j)
    {
        const int s = t0 + j;
        const bool valid = s < nlay;
        const int sc = min(s, nlay-1);
        const int ml = top_at_1 ? sc : nlay-1-sc;
        const Vec<F,V> tv = load_cols<F,V>(tau + lay_base + size_t(ml)*ncl);
        const Vec<F,V> ls = load_cols<F,V>(lay_source + lay_base + size_t(ml)*ncl);

        #pragma unroll
        for (int v=0; v<V; ++v)
        {
            // level source below this layer: next register, or the first level of the next level-lane
            F lev_below;
            if (j < K-1) lev_below = lv[j+1].v[v];
            else         lev_below = lv_next.v[v];
            const F lev_above = lv[j].v[v];

            // a padding layer (level slot beyond the surface) is made transparent through its optical depth: tau = 0 gives
            // trans = 1 and fact = 0 (series branch) exactly, hence zero sources -- one select instead of three
            const F tau_loc = (valid ? tv.v[v] : F(0.)) * D.v[v];
            const F trans = exp_neg(-tau_loc);
            const F fact = tau_loc > tau_thres ?
                (F(1.) - trans) * fast_rcp(tau_loc) - trans :
                tau_loc * (F(.5) + tau_loc * (F(-1./3.) + tau_loc * F(1./8.)));
            const F omt = F(1.) - trans;
            const F s_dn = omt * lev_below + F(2.) * fact * (ls.v[v] - lev_below);
            const F s_up = omt * lev_above + F(2.) * fact * (ls.v[v] - lev_above);

            tr[j][v]  = trans;
            sdn[j][v] = s_dn;
            sup[j][v] = s_up;

            Bdn[v] = tr[j][v] * Bdn[v] + sdn[j][v];
            Bup[v] += A[v] * sup[j][v];
            A[v] *= tr[j][v];
        }
    }

    const Vec<F,V> emis = load_cols<F,V>(sfc_emis + sfc_idx);
    const Vec<F,V> ssrc = load_cols<F,V>(sfc_src + sfc_idx);
    Vec<F,V> inc;
    if (inc_flux != nullptr) inc = load_cols<F,V>(inc_flux + sfc_idx);
    Vec<F,V> sjac;
    if constexpr (JAC) sjac = load_cols<F,V>(sfc_src_jac + sfc_idx);

    F dn_in[V], up_in[V], jac_in[V];

    #pragma unroll
    for (int v=0; v<V; ++v)
    {
        // ---- downward: inclusive scan over level-lanes 0..ll
        F a = A[v], b = Bdn[v];
        #pragma unroll
        for (int d=1; d<LL; d<<=1)
        {
            const F a2 = shfl(a, lane - d*CL);
            const F b2 = shfl(b, lane - d*CL);
            if (ll >= d) { b = a*b2 + b; a = a*a2; }
        }
        F xa = F(1.), xb = F(0.);                               // composite of the levels above this wave's
        F fa = shfl(a, (LL-1)*CL + cl), fb = shfl(b, (LL-1)*CL + cl);   // composite of the whole column
        if constexpr (W >= 2)
        {
            // every wave publishes the composite of its part; (xa, xb) = the parts above this wave's, TOA side first;
            // (fa, fb) = all parts, composed in the same order by every wave (bit-identical dn_sfc in all of them)
            if (ll == LL-1) { xch[4*v+0][wave][cl] = a; xch[4*v+1][wave][cl] = b; }
            __syncthreads();
            const int w0 = wave - h;                     // first wave of this column group
            fa = F(1.); fb = F(0.);
            #pragma unroll
            for (int w=0; w<W; ++w)
            {
                const F oa = xch[4*v+0][w0+w][cl], ob = xch[4*v+1][w0+w][cl];
                if (w == h) { xa = fa; xb = fb; }
                fb = oa*fb + ob; fa = oa*fa;
            }
            if (h > 0) { b = a*xb + b; a = a*xa; }
        }
        F ae = shfl(a, lane - CL), be = shfl(b, lane - CL);     // exclusive
        if (ll == 0) { ae = xa; be = xb; }
        const F dn_top = (inc_flux != nullptr) ? inc.v[v] / pi : F(0.);
        dn_in[v] = ae*dn_top + be;
        const F dn_sfc = fa*dn_top + fb;

        // ---- surface
        const F up_sfc = dn_sfc * (F(1.) - emis.v[v]) + emis.v[v] * ssrc.v[v];

        // ---- upward: inclusive suffix scan over level-lanes ll..7
        a = A[v]; b = Bup[v];
        #pragma unroll
        for (int d=1; d<LL; d<<=1)
        {
            const F a2 = shfl(a, lane + d*CL);
            const F b2 = shfl(b, lane + d*CL);
            if (ll + d < LL) { b = a*b2 + b; a = a*a2; }
        }
        xa = F(1.); xb = F(0.);                                 // composite of the levels below this wave's
        if constexpr (W >= 2)
        {
            if (ll == 0) { xch[4*v+2][wave][cl] = a; xch[4*v+3][wave][cl] = b; }
            __syncthreads();
            const int w0 = wave - h;
            #pragma unroll
            for (int w=W-1; w>=1; --w)                   // the parts below this wave's, surface side first
            {
                if (w > h)
                {
                    const F oa = xch[4*v+2][w0+w][cl], ob = xch[4*v+3][w0+w][cl];
                    xb = oa*xb + ob; xa = oa*xa;
                }
            }
            if (h < W-1) { b = a*xb + b; a = a*xa; }
        }
        ae = shfl(a, lane + CL); be = shfl(b, lane + CL);
        if (ll == LL-1) { ae = xa; be = xb; }
        up_in[v] = ae*up_sfc + be;
        if constexpr (JAC) jac_in[v] = ae * emis.v[v] * sjac.v[v];
    }

    // ---- replay this lane's K layers and store its K levels (each value is stored as soon as it exists: no staging)
    const F scale = pi * w;

    auto put = [&](F* __restrict__ arr, const int j, Vec<F,V> val)
    {
        const int t = t0 + j;
        if (writer && t <= nlay)
        {
            const int ml = top_at_1 ? t : nlay - t;
            F* o = arr + lev_base + size_t(ml)*ncl;
            if constexpr (ACC)
            {
                const Vec<F,V> prev = load_cols<F,V>(o);
                #pragma unroll
                for (int v=0; v<V; ++v) val.v[v] += prev.v[v];
            }
            store_cols<F,V>(o, val);
        }
    };

    {
        F dn[V];
        #pragma unroll
        for (int v=0; v<V; ++v) dn[v] = dn_in[v];
        #pragma unroll
        for (int j=0; j<K; ++j)
        {
            Vec<F,V> o;
            #pragma unroll
            for (int v=0; v<V; ++v) { o.v[v] = scale * dn[v]; dn[v] = tr[j][v]*dn[v] + sdn[j][v]; }
            if constexpr (BB)
            {
                #pragma unroll
                for (int v=0; v<V; ++v) add_rounded(acc_dn[j][v], o.v[v]);
            }
            else put(flux_dn, j, o);
        }
    }
    {
        F up[V], jc[V];
        #pragma unroll
        for (int v=0; v<V; ++v) { up[v] = up_in[v]; jc[v] = JAC ? jac_in[v] : F(0.); }
        #pragma unroll
        for (int j=K-1; j>=0; --j)
        {
            Vec<F,V> o, oj;
            #pragma unroll
            for (int v=0; v<V; ++v)
            {
                up[v] = tr[j][v]*up[v] + sup[j][v];
                o.v[v] = scale * up[v];
                if constexpr (JAC) { jc[v] = tr[j][v]*jc[v]; oj.v[v] = scale * jc[v]; }
            }
            if constexpr (BB)
            {
                #pragma unroll
                for (int v=0; v<V; ++v) add_rounded(acc_up[j][v], o.v[v]);
            }
            else put(flux_up, j, o);
            if constexpr (JAC) put(flux_up_jac, j, oj);
        }
    }
    }   // g-point loop

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
                const size_t o = size_t(icol) + size_t(ml)*ncl;
                Vec<F,V> u, d;
                #pragma unroll
                for (int v=0; v<V; ++v) { u.v[v] = acc_up[j][v]; d.v[v] = acc_dn[j][v]; }
                store_cols<F,V>(flux_up + o, u);
                store_cols<F,V>(flux_dn + o, d);
            }
        }
    }
}



// ---------------------------------------------------------------------------------------------------------------------
// Fused broadband form, second generation (round 2). Same tiling and scans as lw_noscat_scan_kernel<..., BB = true>, plus
//   PRE  : software pipeline over the g-point loop -- the loads of g-point g+1 are requested behind the first scan barrier
//          of g-point g and land during its scans and replays (tools/labs/lw_lab.hip: 2.78 -> 2.59 ms at C4 fp64);
//   LITE : "Planck-lite" inputs. Instead of lay_source and lev_source the kernel reads the Planck fractions pfrac(col,lay,gpt)
//          and the band-integrated Planck functions B_lay(col,lay,bnd), B_lev(col,lev,bnd), and rebuilds
//          lay_source = pfrac*B_lay, lev_source = sqrt(pfrac*pfrac')*B_lev (first and last level: pfrac*B_lev) itself,
//          exactly the expressions of Planck_source_kernel (gas_optics_rrtmgp_kernels.cu:196-314). Two cell arrays read per
//          g-point instead of three, and the Planck kernel writes one instead of two (LW chain at C4: 9.9 -> 7.5 ms).
//          The band's B values sit in per-thread LDS columns and are refreshed when the band changes.
// One quadrature angle, no Jacobian (the general kernel above keeps those).
#ifndef RRX_LW_LACC
#define RRX_LW_LACC 1
#endif
#ifndef RRX_LW_EXP_TABLE
#define RRX_LW_EXP_TABLE 1
#endif
#ifndef RRX_LW_TIMING
#define RRX_LW_TIMING 0   // diagnostic build (tools/sw_timing.sh): every wavefront adds the clocks it spends per phase of a g-point to g_lw_clk
#endif
#if RRX_LW_TIMING
__device__ unsigned long long g_lw_clk[16][8];
#define RRX_LW_T(k) { const unsigned long long t_ = __builtin_readcyclecounter(); lw_acc[k] += t_ - lw_t; lw_t = t_; }
#else
#define RRX_LW_T(k)
#endif
#ifndef RRX_LW_LACC32
#define RRX_LW_LACC32 1    // fp32, two columns per lane: g-point sums in LDS columns too (round 4: the register form spilled 37-41 VGPRs)
#endif
#ifndef RRX_LW_EV
#define RRX_LW_EV 1         // layers of evaluations the scheduler may interleave (2: the same speed, but the fp32 two-column form then keeps 16 B of scratch per lane)
#endif
// NW = wavefronts per workgroup: 4, or 8 with W = 8 for columns of up to 287 layers (round 3: eight waves x four level-lanes x
// nine layers; one workgroup per CU then, the same two waves per SIMD).
// Round 4, fp32: one column per lane with W = 4 and NW = 8 (two column groups per workgroup share each 128-B line of the 64-B
// rows) -- the geometry of the fp32 SW solver (rrx_solver_sw.hip, CLT note); here it serves odd column counts only.
#ifndef RRX_LW_F32_WAVES
#define RRX_LW_F32_WAVES 2
#endif
template<typename F, int V, int K, int W, int CLT, bool LITE, bool PRE, bool GS = false, int EV = RRX_LW_EV, int NW = (W > 4 ? W : 4)>
__global__ void __launch_bounds__(64*NW, (NW > W) ? RRX_LW_F32_WAVES : (NW > 4 ? 1 : 2))
lw_noscat_bb_kernel(
        const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* __restrict__ secants, const F* __restrict__ weights,
        const F* __restrict__ tau, const F* __restrict__ lay_source /* or pfrac */, const F* __restrict__ lev_source,
        const F* __restrict__ blay, const F* __restrict__ blev, const int* __restrict__ gpoint_bands,
        const F* __restrict__ sfc_emis, const F* __restrict__ sfc_src, const F* __restrict__ inc_flux,
        F* __restrict__ flux_up, F* __restrict__ flux_dn, const int gper, const size_t part_stride)
{
    // GS: blockIdx.y = g-point range [g_lo, g_hi) of this workgroup; its sums go to partial array blockIdx.y
    const int g_lo = GS ? blockIdx.y*gper : 0, g_hi = GS ? min(ngpt, g_lo + gper) : ngpt;
    if constexpr (GS) { flux_up += blockIdx.y*part_stride; flux_dn += blockIdx.y*part_stride; }
    constexpr int CL = CLT, LL = 64/CLT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & (CL-1), ll = lane / CL;
    const int h = wave % W, w0 = wave - h;
    // (a workgroup whose row segment is half a 128-B line: the other half belongs to the next workgroup -- rrx::xcd_contiguous)
    const int bx = ((NW/W)*CL*V*sizeof(F) < 128) ? xcd_contiguous(blockIdx.x, gridDim.x) : int(blockIdx.x);
    const int wave_col0 = (bx*(NW/W) + wave/W) * (CL*V);
    __shared__ F xch[4*V][NW][CL];
    __shared__ F lds_b[LITE ? (2*K+1)*V : 1][64*NW];     // per-thread columns: B_lay[K], B_lev[K+1] of the current band
    constexpr bool ETAB = sizeof(F) == 8 && RRX_LW_EXP_TABLE;
    __shared__ F lds_etab[ETAB ? 64 : 1];
    if constexpr (ETAB) { exp_table_fill(lds_etab); __syncthreads(); }
    int icol = wave_col0 + cl*V;
    const bool active = icol < ncol;
    if (!active) icol = (wave_col0 < ncol) ? wave_col0 : 0;
    const bool writer = active && wave_col0 < ncol;
    const int nlev = nlay + 1;
    const size_t ncl = size_t(ncol);
    const int t0 = (h*LL + ll)*K;
    const F pi = F(3.14159265358979323846);
    const F tau_thres = sqrt(sqrt(Lim<F>::eps()));

    // g-point sums of the lane's K levels: in LDS columns for fp64 (the 4*K registers they would take push the kernel past 256
    // VGPRs into scratch; LDS has room for them at the two workgroups per CU the registers allow), in registers for fp32
    constexpr bool LACC = RRX_LW_LACC && ((sizeof(F) == 8 && V == 1) || (sizeof(F) == 4 && V == 2 && RRX_LW_LACC32));
    __shared__ F lds_acc[LACC ? 2*K*V : 1][64*NW];
    F acc_up[LACC ? 1 : K][V], acc_dn[LACC ? 1 : K][V];
    #pragma unroll
    for (int j=0; j<K; ++j)
        #pragma unroll
        for (int v=0; v<V; ++v)
        {
            if constexpr (LACC) { lds_acc[j*V+v][tid] = F(0.); lds_acc[(K+j)*V+v][tid] = F(0.); }
            else { acc_up[j][v] = F(0.); acc_dn[j][v] = F(0.); }
        }

    // element offsets inside one g-point slab: sweep layer s = t0+j, sweep level t = t0+j (clamped into the domain)
    auto lay_off = [&](const int j) -> unsigned
    {
        const int sc = min(max(t0 + j, 0), nlay-1);
        return unsigned(top_at_1 ? sc : nlay-1-sc)*unsigned(ncol) + unsigned(icol);
    };
    auto lev_off = [&](const int j) -> unsigned
    {
        const int tc = min(t0 + j, nlay);
        return unsigned(top_at_1 ? tc : nlay-tc)*unsigned(ncol) + unsigned(icol);
    };

    struct Loads { Vec<F,V> a0[K], a1[K], a2[LITE ? 1 : K], x_next, x_prev, emis, ssrc, D, inc; };
    auto issue = [&](const int g, Loads& L)
    {
        const F* __restrict__ t_g = tau + size_t(g)*ncl*nlay;
        const F* __restrict__ l_g = lay_source + size_t(g)*ncl*nlay;
        #pragma unroll
        for (int j=0; j<K; ++j) { const unsigned o = lay_off(j); L.a0[j] = load_cols<F,V>(t_g + o); L.a1[j] = load_cols<F,V>(l_g + o); }
        if constexpr (!LITE)
        {
            const F* __restrict__ v_g = lev_source + size_t(g)*ncl*nlev;
            #pragma unroll
            for (int j=0; j<K; ++j) L.a2[j] = load_cols<F,V>(v_g + lev_off(j));
            L.x_next = load_cols<F,V>(v_g + lev_off(K));          // level below the lane's last layer
        }
        else
        {
            L.x_next = load_cols<F,V>(l_g + lay_off(K));          // pfrac of the layer below the lane's last one
            L.x_prev = load_cols<F,V>(l_g + lay_off(-1));         // pfrac of the layer above the lane's first one
        }
        const size_t sfc = size_t(g)*ncl + icol;
        L.emis = load_cols<F,V>(sfc_emis + sfc); L.ssrc = load_cols<F,V>(sfc_src + sfc); L.D = load_cols<F,V>(secants + sfc);
        if (inc_flux != nullptr) L.inc = load_cols<F,V>(inc_flux + sfc);
    };

    Loads nxt;
    if constexpr (PRE) issue(g_lo, nxt);
    int cur_bnd = -1;
    const F wgt = weights[0];
    const F scale = pi * wgt;

#if RRX_LW_TIMING
    unsigned long long lw_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lw_t = __builtin_readcyclecounter();
#endif
    for (int igpt=g_lo; igpt<g_hi; ++igpt)
    {
    RRX_LW_T(7)
    if constexpr (!PRE) __syncthreads();        // partner waves issue their load bursts together
    Loads cur;
    if constexpr (PRE) cur = nxt; else issue(igpt, cur);

    if constexpr (LITE)
    {
        const int ib = gpoint_bands[igpt] - 1;                  // wave-uniform
        if (ib != cur_bnd)
        {
            cur_bnd = ib;
            const F* __restrict__ bl = blay + size_t(ib)*ncl*nlay;
            const F* __restrict__ bv = blev + size_t(ib)*ncl*nlev;
            #pragma unroll
            for (int j=0; j<K; ++j)
            {
                const Vec<F,V> x = load_cols<F,V>(bl + lay_off(j));
                #pragma unroll
                for (int v=0; v<V; ++v) lds_b[j*V+v][tid] = x.v[v];
            }
            #pragma unroll
            for (int j=0; j<=K; ++j)
            {
                const Vec<F,V> x = load_cols<F,V>(bv + lev_off(j));
                #pragma unroll
                for (int v=0; v<V; ++v) lds_b[(K+j)*V+v][tid] = x.v[v];
            }
        }
    }

    // level source at sweep level t0+j for column v
    auto level_src = [&](const int j, const int v) -> F
    {
        if constexpr (!LITE) return (j < K) ? cur.a2[min(j, K-1)].v[v] : cur.x_next.v[v];
        else
        {
            const F pa = (j == 0) ? cur.x_prev.v[v] : cur.a1[max(j-1, 0)].v[v];
            const F pb = (j == K) ? cur.x_next.v[v] : cur.a1[min(j, K-1)].v[v];
            const F bvv = lds_b[(K+j)*V+v][tid];
            // The first and the last level take the fraction of their one layer (gas_optics_rrtmgp_kernels.cu:260-306). No select for
            // that (round 4; rounds 2-3 spent four v_cndmask per level on it): the loads of the neighbouring layer are clamped into the
            // column (lay_off), so there pa == pb and sqrt_pos(p*p) returns p -- the residual of its last Newton step is exact.
            return sqrt_pos(pa*pb) * bvv;
        }
    };

    F tr[K][V], sdn[K][V], sup[K][V];
    F A[V], Bdn[V], Bup[V], lva[V];
    #pragma unroll
    for (int v=0; v<V; ++v) { A[v] = F(1.); Bdn[v] = F(0.); Bup[v] = F(0.); lva[v] = level_src(0, v); }
    #pragma unroll
    for (int j=0; j<K; ++j)
    {
        const bool valid = (t0 + j) < nlay;
        #pragma unroll
        for (int v=0; v<V; ++v)
        {
            F tvj = cur.a0[j].v[v];
            const int e = j*V + v - EV*V;                      // at most EV layers of evaluations in flight
            if (e >= 0) asm volatile("" : "+v"(tvj) : "v"(sup[e / V][e % V]));
            const F lvb = level_src(j+1, v);
            F lsj = cur.a1[j].v[v];
            if constexpr (LITE) lsj *= lds_b[j*V+v][tid];
            const F tau_loc = (valid ? tvj : F(0.)) * cur.D.v[v];      // padding layer: tau = 0 -> trans = 1, fact = 0, sources 0 (exactly)
            F trans;
            if constexpr (ETAB) trans = exp_neg(-tau_loc, lds_etab); else trans = exp_neg(-tau_loc);
            const F fact = tau_loc > tau_thres ? (F(1.) - trans) * fast_rcp(tau_loc) - trans
                                               : tau_loc * (F(.5) + tau_loc * (F(-1./3.) + tau_loc * F(1./8.)));
            const F omt = F(1.) - trans;
            const F s_dn = omt * lvb + F(2.) * fact * (lsj - lvb);
            const F s_up = omt * lva[v] + F(2.) * fact * (lsj - lva[v]);
            lva[v] = lvb;
            tr[j][v] = trans; sdn[j][v] = s_dn; sup[j][v] = s_up;
            Bdn[v] = tr[j][v]*Bdn[v] + sdn[j][v];
            Bup[v] += A[v]*sup[j][v];
            A[v] *= tr[j][v];
        }
    }

    RRX_LW_T(0)
    F dn_in[V], up_in[V];
    #pragma unroll
    for (int v=0; v<V; ++v)
    {
        // ---- downward: inclusive scan over the level-lanes, then over the waves of the column group
        F a = A[v], b = Bdn[v];
        #pragma unroll
        for (int d=1; d<LL; d<<=1)
        {
            const F a2 = shfl(a, lane - d*CL), b2 = shfl(b, lane - d*CL);
            if (ll >= d) { b = a*b2 + b; a = a*a2; }
        }
        F xa = F(1.), xb = F(0.);
        if (ll == LL-1) { xch[4*v+0][wave][cl] = a; xch[4*v+1][wave][cl] = b; }
        RRX_LW_T(1)
        __syncthreads();
        RRX_LW_T(6)
        if constexpr (PRE)
        {
            if (v == 0)
            {
                // every wave of the workgroup is here: the waves that share 128-B lines ask for them together
                __builtin_amdgcn_sched_barrier(0);
                issue(min(igpt + 1, g_hi - 1), nxt);            // (last iteration: a harmless re-read)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        F fa = F(1.), fb = F(0.);
        #pragma unroll
        for (int w=0; w<W; ++w)
        {
            const F oa = xch[4*v+0][w0+w][cl], ob = xch[4*v+1][w0+w][cl];
            if (w == h) { xa = fa; xb = fb; }
            fb = oa*fb + ob; fa = oa*fa;
        }
        if (h > 0) { b = a*xb + b; a = a*xa; }
        F ae = shfl(a, lane - CL), be = shfl(b, lane - CL);
        if (ll == 0) { ae = xa; be = xb; }
        const F dn_top = (inc_flux != nullptr) ? cur.inc.v[v] / pi : F(0.);
        dn_in[v] = ae*dn_top + be;
        const F dn_sfc = fa*dn_top + fb;
        const F up_sfc = dn_sfc * (F(1.) - cur.emis.v[v]) + cur.emis.v[v] * cur.ssrc.v[v];

        // ---- upward: inclusive suffix scan
        a = A[v]; b = Bup[v];
        #pragma unroll
        for (int d=1; d<LL; d<<=1)
        {
            const F a2 = shfl(a, lane + d*CL), b2 = shfl(b, lane + d*CL);
            if (ll + d < LL) { b = a*b2 + b; a = a*a2; }
        }
        xa = F(1.); xb = F(0.);
        if (ll == 0) { xch[4*v+2][wave][cl] = a; xch[4*v+3][wave][cl] = b; }
        RRX_LW_T(2)
        __syncthreads();
        RRX_LW_T(6)
        #pragma unroll
        for (int w=W-1; w>=1; --w)
            if (w > h) { const F oa = xch[4*v+2][w0+w][cl], ob = xch[4*v+3][w0+w][cl]; xb = oa*xb + ob; xa = oa*xa; }
        if (h < W-1) { b = a*xb + b; a = a*xa; }
        ae = shfl(a, lane + CL); be = shfl(b, lane + CL);
        if (ll == LL-1) { ae = xa; be = xb; }
        up_in[v] = ae*up_sfc + be;
    }

    #pragma unroll
    for (int v=0; v<V; ++v)
    {
        F dn = dn_in[v];
        #pragma unroll
        for (int j=0; j<K; ++j)
        {
            if constexpr (LACC) { F a = lds_acc[(K+j)*V+v][tid]; add_rounded(a, scale*dn); lds_acc[(K+j)*V+v][tid] = a; }
            else add_rounded(acc_dn[j][v], scale*dn);
            dn = tr[j][v]*dn + sdn[j][v];
        }
        F up = up_in[v];
        #pragma unroll
        for (int j=K-1; j>=0; --j)
        {
            up = tr[j][v]*up + sup[j][v];
            if constexpr (LACC) { F a = lds_acc[j*V+v][tid]; add_rounded(a, scale*up); lds_acc[j*V+v][tid] = a; }
            else add_rounded(acc_up[j][v], scale*up);
        }
    }
    RRX_LW_T(3)
    }   // g-point loop
#if RRX_LW_TIMING
    if (lane == 0) for (int k=0; k<8; ++k) atomicAdd(&g_lw_clk[wave & 15][k], lw_acc[k]);
#endif

    if (!writer) return;
    #pragma unroll
    for (int j=0; j<K; ++j)
    {
        const int t = t0 + j;
        if (t <= nlay)
        {
            const size_t o = size_t(icol) + size_t(top_at_1 ? t : nlay - t)*ncl;
            Vec<F,V> u, d;
            #pragma unroll
            for (int v=0; v<V; ++v)
            {
                if constexpr (LACC) { u.v[v] = lds_acc[j*V+v][tid]; d.v[v] = lds_acc[(K+j)*V+v][tid]; }
                else { u.v[v] = acc_up[j][v]; d.v[v] = acc_dn[j][v]; }
            }
            store_cols<F,V>(flux_up + o, u);
            store_cols<F,V>(flux_dn + o, d);
        }
    }
}

// Any-nlay fallback: one thread per (col, gpt), layer quantities recomputed in the second sweep
// (no scratch). Used when nlay+1 > 8*K_MAX, and as the A/B baseline in bench.py --variant serial.
template<typename F, bool JAC, bool ACC>
__global__ void __launch_bounds__(256)
lw_noscat_serial_kernel(
        const int ncol, const int nlay, const int ngpt, const int top_at_1, const int imu,
        const F* __restrict__ secants, const F* __restrict__ weights,
        const F* __restrict__ tau, const F* __restrict__ lay_source, const F* __restrict__ lev_source,
        const F* __restrict__ sfc_emis, const F* __restrict__ sfc_src, const F* __restrict__ inc_flux,
        F* __restrict__ flux_up, F* __restrict__ flux_dn,
        const F* __restrict__ sfc_src_jac, F* __restrict__ flux_up_jac)
{
    const int icol = blockIdx.x*blockDim.x + threadIdx.x;
    const int igpt = blockIdx.y;
    if (icol >= ncol) return;

    const int nlev = nlay+1;
    const size_t ncl = size_t(ncol);
    const size_t lay_base = size_t(igpt)*ncl*nlay + icol;
    const size_t lev_base = size_t(igpt)*ncl*nlev + icol;
    const size_t sfc_idx = size_t(igpt)*ncl + icol;
    const F pi = F(3.14159265358979323846);
    const F tau_thres = sqrt(sqrt(Lim<F>::eps()));
    const F D = secants[sfc_idx + size_t(imu)*ncl*ngpt];
    const F scale = pi * weights[imu];

    auto layer = [&](const int s, F& trans, F& s_dn, F& s_up)
    {
        const int ml = top_at_1 ? s : nlay-1-s;
        const int m_above = top_at_1 ? ml : ml+1;
        const int m_below = top_at_1 ? ml+1 : ml;
        const F tau_loc = tau[lay_base + size_t(ml)*ncl] * D;
        const F ls = lay_source[lay_base + size_t(ml)*ncl];
        const F lev_above = lev_source[lev_base + size_t(m_above)*ncl];
        const F lev_below = lev_source[lev_base + size_t(m_below)*ncl];
        trans = exp(-tau_loc);
        const F fact = tau_loc > tau_thres ?
            (F(1.) - trans) / tau_loc - trans :
            tau_loc * (F(.5) + tau_loc * (F(-1./3.) + tau_loc * F(1./8.)));
        s_dn = (F(1.) - trans) * lev_below + F(2.) * fact * (ls - lev_below);
        s_up = (F(1.) - trans) * lev_above + F(2.) * fact * (ls - lev_above);
    };
    auto put = [&](F* arr, const int t, const F val)
    {
        const int ml = top_at_1 ? t : nlay - t;
        const size_t o = lev_base + size_t(ml)*ncl;
        arr[o] = ACC ? arr[o] + scale*val : scale*val;
    };

    F dn = (inc_flux != nullptr) ? inc_flux[sfc_idx] / pi : F(0.);
    for (int s=0; s<nlay; ++s)
    {
        F trans, s_dn, s_up;
        layer(s, trans, s_dn, s_up);
        put(flux_dn, s, dn);
        dn = trans*dn + s_dn;
    }
    put(flux_dn, nlay, dn);

    const F emis = sfc_emis[sfc_idx];
    F up = dn * (F(1.) - emis) + emis * sfc_src[sfc_idx];
    F jc = JAC ? emis * sfc_src_jac[sfc_idx] : F(0.);
    put(flux_up, nlay, up);
    if constexpr (JAC) put(flux_up_jac, nlay, jc);
    for (int s=nlay-1; s>=0; --s)
    {
        F trans, s_dn, s_up;
        layer(s, trans, s_dn, s_up);
        up = trans*up + s_up;
        put(flux_up, s, up);
        if constexpr (JAC) { jc = trans*jc; put(flux_up_jac, s, jc); }
    }
}


template<typename F>
__global__ void lw_secants_array_kernel(
        const int ncol, const int ngpt, const int n_gauss_quad, const int max_gauss_pts,
        const F* __restrict__ gauss_Ds, F* __restrict__ secants)
{
    const size_t n = size_t(ncol)*ngpt*n_gauss_quad;
    for (size_t i = size_t(blockIdx.x)*blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x)*blockDim.x)
    {
        const int imu = int(i / (size_t(ncol)*ngpt));
        secants[i] = gauss_Ds[imu + (n_gauss_quad-1)*max_gauss_pts];
    }
}


// level array (ncol,nlay+1) += / = sum over g-points of a (ncol,nlay+1,ngpt) array; used for do_broadband
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


template<typename F, int V, int K, int W>
void launch_scan_k(
        hipStream_t st, const dim3 grid, const bool jac, const bool acc,
        const int ncol, const int nlay, const int ngpt, const int top_at_1, const int imu,
        const F* secants, const F* weights, const F* tau, const F* lay_source, const F* lev_source,
        const F* sfc_emis, const F* sfc_src, const F* inc_flux, F* flux_up, F* flux_dn,
        const F* sfc_src_jac, F* flux_up_jac)
{
#define RRX_LW_ARGS ncol, nlay, ngpt, top_at_1, imu, secants, weights, tau, lay_source, lev_source, \
        sfc_emis, sfc_src, inc_flux, flux_up, flux_dn, sfc_src_jac, flux_up_jac
#define RRX_LW_KARGS RRX_LW_ARGS, tuning().sync_waves
    if (jac)
    {
        if (acc) lw_noscat_scan_kernel<F,V,K,W,true,true><<<grid, 256, 0, st>>>(RRX_LW_KARGS);
        else     lw_noscat_scan_kernel<F,V,K,W,true,false><<<grid, 256, 0, st>>>(RRX_LW_KARGS);
    }
    else
    {
        if (acc) lw_noscat_scan_kernel<F,V,K,W,false,true><<<grid, 256, 0, st>>>(RRX_LW_KARGS);
        else     lw_noscat_scan_kernel<F,V,K,W,false,false><<<grid, 256, 0, st>>>(RRX_LW_KARGS);
    }
}

template<typename F, int V, int W>
bool launch_scan(
        hipStream_t st, const bool jac, const bool acc,
        const int ncol, const int nlay, const int ngpt, const int top_at_1, const int imu,
        const F* secants, const F* weights, const F* tau, const F* lay_source, const F* lev_source,
        const F* sfc_emis, const F* sfc_src, const F* inc_flux, F* flux_up, F* flux_dn,
        const F* sfc_src_jac, F* flux_up_jac)
{
    const dim3 grid(ceil_div(ncol, (4/W)*CL*V), ngpt);
    const int need = ceil_div(nlay+1, LL*W);
#define RRX_LW_K(KK) if (need <= KK) { launch_scan_k<F,V,KK,W>(st, grid, jac, acc, RRX_LW_ARGS); return true; }
    if constexpr (W == 1)      { RRX_LW_K(4) RRX_LW_K(8) RRX_LW_K(12) RRX_LW_K(18) RRX_LW_K(24) RRX_LW_K(33) }
    else if constexpr (W == 2) { RRX_LW_K(2) RRX_LW_K(4) RRX_LW_K(6)  RRX_LW_K(9)  RRX_LW_K(12) RRX_LW_K(17) }
    else                       { RRX_LW_K(2) RRX_LW_K(3) RRX_LW_K(5) }
#undef RRX_LW_K
    return false;
}

// 0 = default, 1 = serial fallback, 2 = one wave/V=1, 3 = one wave/wide rows, 4 = two waves/64-B rows,
// 5 = two waves/128-B rows, 6 = one wave/64-B rows, 7 = default kernels but never the fused broadband form,
// 10 = four waves/128-B rows (per-g-point form), 8 / 9 / 12 = fused broadband form with two / four waves of 8 x 8 lanes / four waves of 16 x 4 lanes
// (default: 12 in fp64, 9 in fp32)
// fused broadband form: one workgroup walks all g-points of its columns (grid.y = 1)
template<typename F, int V, int W>
bool launch_scan_bb(
        hipStream_t st, const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* secants, const F* weights, const F* tau, const F* lay_source, const F* lev_source,
        const F* sfc_emis, const F* sfc_src, const F* inc_flux, F* flux_up, F* flux_dn)
{
    const dim3 grid(ceil_div(ncol, (4/W)*CL*V), 1);
    const int need = ceil_div(nlay+1, LL*W);
    const int imu = 0;
    const F* sfc_src_jac = nullptr; F* flux_up_jac = nullptr;
#define RRX_LW_K(KK) if (need <= KK) { lw_noscat_scan_kernel<F,V,KK,W,false,false,true><<<grid, 256, 0, st>>>(RRX_LW_KARGS); return true; }
    if constexpr (W == 1)      { RRX_LW_K(4) RRX_LW_K(8) RRX_LW_K(12) RRX_LW_K(18) }
    else if constexpr (W == 2) { RRX_LW_K(2) RRX_LW_K(4) RRX_LW_K(6)  RRX_LW_K(9)  RRX_LW_K(12) }
    else                       { RRX_LW_K(2) RRX_LW_K(3) RRX_LW_K(5) }     // taller columns spill at 128-B rows: W = 2 form
#undef RRX_LW_K
    return false;
}

// fused broadband form in the 16 column-lane x 4 level-lane geometry, four waves per column group
template<typename F, int V>
bool launch_scan_bb16(
        hipStream_t st, const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* secants, const F* weights, const F* tau, const F* lay_source, const F* lev_source,
        const F* sfc_emis, const F* sfc_src, const F* inc_flux, F* flux_up, F* flux_dn)
{
    const dim3 grid(ceil_div(ncol, 16*V), 1);
    const int need = ceil_div(nlay+1, 4*4);
    const int imu = 0;
    const F* sfc_src_jac = nullptr; F* flux_up_jac = nullptr;
#define RRX_LW_K(KK) if (need <= KK) { lw_noscat_scan_kernel<F,V,KK,4,false,false,true,16><<<grid, 256, 0, st>>>(RRX_LW_KARGS); return true; }
    RRX_LW_K(2) RRX_LW_K(4) RRX_LW_K(6) RRX_LW_K(9) RRX_LW_K(12)
#undef RRX_LW_K
    return false;
}



// second-generation fused broadband kernel (lw_noscat_bb_kernel); LITE: lay_source = pfrac, lev_source unused
template<typename F, int V, int W, int CLT, bool LITE, int NW = (W > 4 ? W : 4)>
bool launch_bb2(
        hipStream_t st, const bool pre, const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* secants, const F* weights, const F* tau, const F* lay_source, const F* lev_source,
        const F* blay, const F* blev, const int* gpoint_bands,
        const F* sfc_emis, const F* sfc_src, const F* inc_flux, F* flux_up, F* flux_dn)
{
    if (size_t(ncol)*(nlay+1) >= (size_t(1) << 31)) return false;          // 32-bit element offsets inside a g-point slab
    const int groups = ceil_div(ncol, (NW/W)*CLT*V);
    const int need = ceil_div(nlay+1, (64/CLT)*W);
    if (need > ((CLT == 16 || W == 8) ? 9 : 5)) return false;
    // few column groups: the g-point loop is split over grid.y, partial sums added in range order afterwards
    const int gper = ceil_div(ngpt, broadband_gsplit(groups, ngpt, (NW > 4) ? 256 : 512));      // (one or two workgroups per CU)
    const int nsplit = ceil_div(ngpt, gper);               // no empty range: every workgroup's first g-point exists (it is prefetched)
    const size_t nlevcol = size_t(ncol)*(nlay+1);
    StreamScratch scratch(st);
    F* out_up = flux_up; F* out_dn = flux_dn;
    if (nsplit > 1) { out_up = scratch.get<F>(2*nsplit*nlevcol); out_dn = out_up + nsplit*nlevcol; }
    const dim3 grid(groups, nsplit);
#define RRX_LW_B2(KK) if (need <= KK) { \
        if (nsplit > 1 && pre) lw_noscat_bb_kernel<F,V,KK,W,CLT,LITE,true,true,RRX_LW_EV,NW><<<grid, 64*NW, 0, st>>>(ncol, nlay, ngpt, top_at_1, secants, weights, tau, \
            lay_source, lev_source, blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, out_up, out_dn, gper, nlevcol); \
        else if (nsplit > 1) lw_noscat_bb_kernel<F,V,KK,W,CLT,LITE,false,true,RRX_LW_EV,NW><<<grid, 64*NW, 0, st>>>(ncol, nlay, ngpt, top_at_1, secants, weights, tau, \
            lay_source, lev_source, blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, out_up, out_dn, gper, nlevcol); \
        else if (pre) lw_noscat_bb_kernel<F,V,KK,W,CLT,LITE,true,false,RRX_LW_EV,NW><<<grid, 64*NW, 0, st>>>(ncol, nlay, ngpt, top_at_1, secants, weights, tau, \
            lay_source, lev_source, blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, out_up, out_dn, gper, nlevcol); \
        else lw_noscat_bb_kernel<F,V,KK,W,CLT,LITE,false,false,RRX_LW_EV,NW><<<grid, 64*NW, 0, st>>>(ncol, nlay, ngpt, top_at_1, secants, weights, tau, \
            lay_source, lev_source, blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, out_up, out_dn, gper, nlevcol); \
        break; }
    do {
    if constexpr (CLT == 16) { RRX_LW_B2(2) RRX_LW_B2(4) RRX_LW_B2(6) RRX_LW_B2(9) }
    else if constexpr (W == 8) { RRX_LW_B2(5) RRX_LW_B2(7) RRX_LW_B2(9) }      // (288 ... 319 / 447 / 575 layers: eight waves of 8 x 8 lanes)
    else                     { RRX_LW_B2(2) RRX_LW_B2(3) RRX_LW_B2(5) }
    } while (false);
    if (nsplit > 1)      // (out_up, out_dn lie behind each other in the scratch block)
        sum_ranges_kernel<F,2><<<dim3(ceil_div(nlevcol, 256), 2), 256, 0, st>>>(nlevcol, nsplit, out_up, flux_up, flux_dn, (F*)nullptr);
    return true;
#undef RRX_LW_B2
}

// broadband fluxes from tau + (lay_source, lev_source) [LITE = false] or tau + Planck fractions and band Planck functions
// [LITE = true] in the one-kernel form; false when the shape is outside its tilings (the caller takes another path)
template<typename F, bool LITE>
bool lw_fused_broadband(
        hipStream_t st, const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* secants, const F* weights, const F* tau, const F* lay_source, const F* lev_source,
        const F* blay, const F* blev, const int* gpoint_bands,
        const F* sfc_emis, const F* sfc_src, const F* inc_flux, F* flux_up, F* flux_dn)
{
    const bool pre = tuning().lw_variant != 13;                 // 13: without the pipelined loads (A/B runs)
    // fp32: 16 x 4 lanes with two columns per lane (128-B rows, K = 9) ahead of 8 x 8 lanes with four (variant 14 = the latter
    // first, for A/B runs). Measured at C4 in the fractions form: 1.77 against 3.26 ms (the four-column lane state spills).
    const bool v2_first = tuning().lw_variant != 14;
    if constexpr (sizeof(F) == 8)
    {
        // (Round 4 measured six waves x six layers per column group -- 384-thread workgroups, three waves per SIMD, 168 VGPRs with
        //  108-124 B of scratch: 4.2-4.7 ms against 2.7 for this form, profiles/r04_fp32_geometry_ab.txt.)
        if (launch_bb2<F,1,4,16,LITE>(st, pre, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                      blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn))
            return true;
        // 144 ... 287 layers: eight wavefronts per column group
        if (launch_bb2<F,1,8,16,LITE>(st, pre, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                      blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn))
            return true;
        // 288 ... 575 layers (round 4: RCEMIP's default is 256 levels, LES grids with a background profile on top exceed 288): the same
        // eight waves with 8 x 8 lanes -- 64 levels per wave at nine layers per lane, 64-B rows (the other half of each 128-B line
        // belongs to the next column group: twice the L2 fetches, on a kernel that stands at a quarter of the HBM roof). Beyond that
        // the one-thread-per-column kernels take over.
        return launch_bb2<F,1,8,8,LITE>(st, pre, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                        blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn);
    }
    else
    {
        // one column per lane, two column groups per workgroup, four waves per SIMD (variant 15 = the forms of rounds 1-3)
        // Round 4 measured the one-column-per-lane geometries of the SW solver here too (K = 9 / W = 4 at two waves per SIMD, K = 6 /
        // W = 6 at three, K = 5 / W = 8 at four: 1.82 / 1.88 / 1.95 ms at C4 against 1.31 for two columns per lane with the sums in
        // LDS, profiles/r04_fp32_geometry_ab.txt): the LW chain per g-point is short, so halving the wavefronts per column wins.
        // The one-column form stays for odd column counts (variant 15 forces it for tests).
        if (tuning().lw_variant == 15 &&
            launch_bb2<F,1,4,16,LITE,8>(st, pre, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                        blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn))
            return true;
        if (ncol % 2 == 0 && v2_first &&
            launch_bb2<F,2,4,16,LITE>(st, pre, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                      blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn))
            return true;
        if (ncol % 4 == 0 &&
            launch_bb2<F,4,4,8,LITE>(st, pre, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                     blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn))
            return true;
        if (ncol % 2 == 0 &&
            launch_bb2<F,2,4,16,LITE>(st, pre, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                      blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn))
            return true;
        // 144 ... 287 layers: eight wavefronts per column group
        if (ncol % 2 == 0 &&
            launch_bb2<F,2,8,16,LITE>(st, pre, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                      blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn))
            return true;
        // 288 ... 575 layers: eight waves of 8 x 8 lanes (see fp64)
        if (ncol % 2 == 0 &&
            launch_bb2<F,2,8,8,LITE>(st, pre, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                     blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn))
            return true;
        // odd column counts: one column per lane
        return launch_bb2<F,1,4,16,LITE,8>(st, pre, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                           blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, flux_up, flux_dn);
    }
}

#define RRX_LW_ARGS_CALL ncol, nlay, ngpt, top_at_1, imu, secants, weights, tau, lay_source, lev_source, \
        sfc_emis, sfc_src, inc_flux, up, dn, sfc_src_jac, flux_up_jac

template<typename F>
int lw_solver_noscat_impl(
        const int ncol, const int nlay, const int ngpt, const Bool top_at_1, const int nmus,
        const F* secants, const F* weights,
        const F* tau, const F* lay_source, const F* lev_source,
        const F* sfc_emis, const F* sfc_src, const F* inc_flux,
        F* flux_up, F* flux_dn,
        const Bool do_broadband, F* flux_up_loc, F* flux_dn_loc,
        const Bool do_jacobians, const F* sfc_src_jac, F* flux_up_jac,
        void* stream, F* flux_ws = nullptr /* room for 2*ncol*(nlay+1)*ngpt values out of the CALLER's workspace lease, or null */)
{
    RRX_TRY
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ncol <= 0 || nlay <= 0 || ngpt <= 0) throw std::runtime_error("empty problem");
    if (nmus < 1 || nmus > 4) throw std::runtime_error("n_quad_angs must be 1..4");
    const bool jac = do_jacobians && sfc_src_jac != nullptr && flux_up_jac != nullptr;
    const int g_lw_variant = tuning().lw_variant;
    const int g_bb_min_groups = tuning().bb_min_groups;

    // broadband mode, fused form: g-point sums kept in registers, no per-g-point fluxes in memory. With enough column groups
    // to fill the chip one workgroup sums all g-points in order (sum_broadband's order); with fewer the g-point range is split
    // over grid.y and the partial sums are added in range order (rrx::broadband_gsplit).
    constexpr int VBB = (sizeof(F) == 8) ? 1 : 2;
    const bool second_gen = (g_lw_variant == 0 || (g_lw_variant >= 13 && g_lw_variant <= 15));        // splits its g-point loop when columns are few
    if (do_broadband && !jac && nmus == 1 && g_lw_variant != 1 && g_lw_variant != 7 && (ncol % VBB == 0 || (second_gen && sizeof(F) == 4))
        && (second_gen || ceil_div(ncol, CL*VBB) >= g_bb_min_groups))
    {
        if (flux_up_loc == nullptr || flux_dn_loc == nullptr) throw std::runtime_error("do_broadband needs flux_*_loc");
        // default: the second-generation kernel (pipelined loads); variants 8 / 9 / 12 keep the first-generation tilings
        if (second_gen &&
            lw_fused_broadband<F,false>(st, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                        (const F*)nullptr, (const F*)nullptr, (const int*)nullptr, sfc_emis, sfc_src, inc_flux,
                                        flux_up_loc, flux_dn_loc))
            return 0;
        // the first-generation tilings do not split the g-point loop: only with enough column groups
        if (ncol % VBB == 0 && ceil_div(ncol, CL*VBB) >= g_bb_min_groups) {
        // four waves per column group (K = 5 at 140 layers) leave room for the g-point sums AND 128-B row segments.
        // Measured at C4: fp32 2.15 ms against 2.28 ms with two waves / 64-B rows; fp64 3.94 against 3.15 ms (256 VGPRs,
        // 12 % idle level-lanes), so fp64 keeps two waves unless variant 9 asks for four.
        // fp64 default: 16 x 4 lane geometry over four waves (128-B rows at K = 9: 3.00 -> 2.84 ms at C4; fp32 prefers the
        // 8 x 8 geometry with four waves and V = 4, 2.16 against 2.28 ms)
        if ((g_lw_variant == 12 || (g_lw_variant == 0 && sizeof(F) == 8)) && ncol % VBB == 0 &&
            launch_scan_bb16<F,VBB>(st, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                    sfc_emis, sfc_src, inc_flux, flux_up_loc, flux_dn_loc))
            return 0;
        const bool four = (g_lw_variant == 9) || (g_lw_variant != 8 && sizeof(F) == 4);
        if (four && ncol % (2*VBB) == 0 &&
            launch_scan_bb<F,2*VBB,4>(st, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                      sfc_emis, sfc_src, inc_flux, flux_up_loc, flux_dn_loc))
            return 0;
        if (launch_scan_bb<F,VBB,2>(st, ncol, nlay, ngpt, top_at_1, secants, weights, tau, lay_source, lev_source,
                                    sfc_emis, sfc_src, inc_flux, flux_up_loc, flux_dn_loc))
            return 0;
        }
    }

    // broadband mode, general form: per-g-point fluxes go to a workspace, then are summed over g-points
    F* up = flux_up; F* dn = flux_dn;
    WorkspaceLease lease(st);
    const size_t nlevcol = size_t(ncol)*(nlay+1);
    if (do_broadband)
    {
        if (flux_up_loc == nullptr || flux_dn_loc == nullptr) throw std::runtime_error("do_broadband needs flux_*_loc");
        F* ws = (flux_ws != nullptr) ? flux_ws : lease.get<F>(2*nlevcol*ngpt);
        up = ws; dn = ws + nlevcol*ngpt;
    }

    // columns per lane: VDEF*8 lanes*sizeof(F) = 64-B row segments, VMAX = 128-B segments. Measured at C4 (tools/
    // bench_solvers.py): two waves per column group with 128-B rows (2 waves/SIMD) is the fastest form in both
    // precisions (fp64 5.4 ms vs 6.0 ms for one wave/64-B rows; fp32 2.4 vs 2.9 ms); the others stay for A/B runs.
    constexpr int VDEF = (sizeof(F) == 8) ? 1 : 2;
    constexpr int VMAX = 2*VDEF;
    for (int imu=0; imu<nmus; ++imu)
    {
        const bool acc = imu > 0;
        bool done = false;
        if (g_lw_variant != 1)
        {
            const int var = (g_lw_variant == 0) ? RRX_LW_DEFAULT_VARIANT : g_lw_variant;
            if (var == 10 && ncol % VMAX == 0)     done = launch_scan<F,VMAX,4>(st, jac, acc, RRX_LW_ARGS_CALL);
            else if (var == 3 && ncol % VMAX == 0) done = launch_scan<F,VMAX,1>(st, jac, acc, RRX_LW_ARGS_CALL);
            else if (var == 6 && ncol % VDEF == 0) done = launch_scan<F,VDEF,1>(st, jac, acc, RRX_LW_ARGS_CALL);
            else if (var == 2)                     done = launch_scan<F,1,1>(st, jac, acc, RRX_LW_ARGS_CALL);
            else if (var != 4 && ncol % VMAX == 0) done = launch_scan<F,VMAX,2>(st, jac, acc, RRX_LW_ARGS_CALL);
            else if (ncol % VDEF == 0)             done = launch_scan<F,VDEF,2>(st, jac, acc, RRX_LW_ARGS_CALL);
            if (!done)                             done = launch_scan<F,1,2>(st, jac, acc, RRX_LW_ARGS_CALL);
        }
        if (!done)
        {
            const dim3 grid(ceil_div(ncol, 256), ngpt);
#define RRX_LW_SERIAL(J, A) lw_noscat_serial_kernel<F,J,A><<<grid, 256, 0, st>>>( \
        ncol, nlay, ngpt, top_at_1, imu, secants, weights, tau, lay_source, lev_source, \
        sfc_emis, sfc_src, inc_flux, up, dn, sfc_src_jac, flux_up_jac)
            if (jac) { if (acc) RRX_LW_SERIAL(true, true); else RRX_LW_SERIAL(true, false); }
            else     { if (acc) RRX_LW_SERIAL(false, true); else RRX_LW_SERIAL(false, false); }
#undef RRX_LW_SERIAL
        }
    }

    if (do_broadband)
    {
        const int nb = ceil_div(nlevcol, 256);
        sum_gpt_kernel<F><<<nb, 256, 0, st>>>(nlevcol, ngpt, up, flux_up_loc);
        sum_gpt_kernel<F><<<nb, 256, 0, st>>>(nlevcol, ngpt, dn, flux_dn_loc);
    }
    RRX_CATCH("rrx_lw_solver_noscat")
}

template<typename F>
__global__ void planck_sources_from_fractions_kernel(
        const int ncol, const int nlay, const int ngpt, const int* __restrict__ gpoint_bands,
        const F* __restrict__ pf, const F* __restrict__ blay, const F* __restrict__ blev,
        F* __restrict__ lay_src, F* __restrict__ lev_src)
{
    // lay_source = pfrac*B_lay; lev_source(level m) = sqrt(pfrac(m)*pfrac(m-1))*B_lev(m), first / last level pfrac*B_lev:
    // the expressions (and rounding) of Planck_source_kernel, gas_optics_rrtmgp_kernels.cu:260-306
    const size_t ncl = ncol; const int nlev = nlay+1;
    const size_t n = ncl*nlev*ngpt;
    for (size_t i = size_t(blockIdx.x)*blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x)*blockDim.x)
    {
        const int icol = int(i % ncl), m = int((i / ncl) % nlev), ig = int(i / (ncl*nlev));
        const int ib = gpoint_bands[ig] - 1;
        const size_t lb = size_t(ig)*ncl*nlay + icol;
        const F bl = blev[(size_t(ib)*nlev + m)*ncl + icol];
        F v;
        if (m == 0) v = pf[lb] * bl;
        else if (m == nlay) v = pf[lb + size_t(nlay-1)*ncl] * bl;
        else v = sqrt(pf[lb + size_t(m)*ncl] * pf[lb + size_t(m-1)*ncl]) * bl;
        lev_src[i] = v;
        if (m < nlay) lay_src[lb + size_t(m)*ncl] = pf[lb + size_t(m)*ncl] * blay[(size_t(ib)*nlay + m)*ncl + icol];
    }
}

template<typename F>
int planck_sources_from_fractions_impl(int ncol, int nlay, int ngpt, const int* gpoint_bands, const F* pfrac, const F* blay,
        const F* blev, F* lay_src, F* lev_src, void* stream)
{
    RRX_TRY
    if (ncol <= 0 || nlay <= 0 || ngpt <= 0) throw std::runtime_error("empty problem");
    const size_t n = size_t(ncol)*(nlay+1)*ngpt;
    planck_sources_from_fractions_kernel<F><<<int(std::min<size_t>((n + 255)/256, 256*16)), 256, 0, static_cast<hipStream_t>(stream)>>>(
            ncol, nlay, ngpt, gpoint_bands, pfrac, blay, blev, lay_src, lev_src);
    RRX_CATCH("rrx_planck_sources_from_fractions")
}

template<typename F>
int lw_solver_noscat_fractions_impl(
        const int ncol, const int nlay, const int ngpt, const Bool top_at_1,
        const F* secants, const F* weights, const F* tau, const F* pfrac, const F* blay, const F* blev, const int* gpoint_bands,
        const F* sfc_emis, const F* sfc_src, const F* inc_flux, F* flux_up_loc, F* flux_dn_loc, void* stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        RRX_TRY
        if (ncol <= 0 || nlay <= 0 || ngpt <= 0) throw std::runtime_error("empty problem");
        if (flux_up_loc == nullptr || flux_dn_loc == nullptr) throw std::runtime_error("broadband outputs missing");
        const int var = tuning().lw_variant;
        if ((var == 0 || (var >= 13 && var <= 15)) &&
            lw_fused_broadband<F,true>(st, ncol, nlay, ngpt, top_at_1, secants, weights, tau, pfrac, (const F*)nullptr,
                                       blay, blev, gpoint_bands, sfc_emis, sfc_src, inc_flux, flux_up_loc, flux_dn_loc))
            return check_launch("rrx_lw_solver_noscat_fractions");
        } catch (const std::exception& e) { rrx::set_error(std::string("rrx_lw_solver_noscat_fractions: ") + e.what()); return 1; }
    }
    // outside the one-kernel form (few columns, very tall columns, A/B variants): rebuild the sources and take the general entry.
    // ONE lease of the stream's workspace, carved here: [per-g-point fluxes of the general entry: 2 n_lev | lay_source | lev_source];
    // the general entry is handed its part explicitly.
    const size_t n_lay = size_t(ncol)*nlay*ngpt, n_lev = size_t(ncol)*(nlay+1)*ngpt;
    WorkspaceLease lease(st);
    F* flux_ws = nullptr;
    try { flux_ws = lease.get<F>(2*n_lev + n_lay + n_lev); }
    catch (const std::exception& e) { rrx::set_error(std::string("rrx_lw_solver_noscat_fractions: ") + e.what()); return 1; }
    F* lay = flux_ws + 2*n_lev; F* lev = lay + n_lay;
    int rc = planck_sources_from_fractions_impl<F>(ncol, nlay, ngpt, gpoint_bands, pfrac, blay, blev, lay, lev, stream);
    if (rc == 0)
        rc = lw_solver_noscat_impl<F>(ncol, nlay, ngpt, top_at_1, 1, secants, weights, tau, lay, lev, sfc_emis, sfc_src, inc_flux,
                                      (F*)nullptr, (F*)nullptr, Bool(1), flux_up_loc, flux_dn_loc, Bool(0), (const F*)nullptr, (F*)nullptr, stream,
                                      flux_ws);
    return rc;
}
}  // namespace


extern "C"
{
int rrx_set_lw_variant(int v) { rrx::tuning().lw_variant = v; return 0; }
#if RRX_LW_TIMING
// diagnostic build only: phase clocks per wavefront of a workgroup (out[16][8]: sources + transmissivities, down scan, up scan, replays + sums,
// -, -, barrier waits, loop top) summed over the workgroups since the last call, then reset
int rrx_lw_timing(unsigned long long* out)
{
    unsigned long long zero[16*8] = {0};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lw_clk), 16*8*sizeof(unsigned long long)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lw_clk), zero, sizeof(zero)) == hipSuccess ? 0 : 1;
}
#endif

int rrx_lw_secants_array_f64(int ncol, int ngpt, int n_gauss_quad, int max_gauss_pts, const double* gauss_Ds, double* secants, void* stream)
{
    RRX_TRY
    lw_secants_array_kernel<double><<<rrx::ceil_div(size_t(ncol)*ngpt*n_gauss_quad, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
            ncol, ngpt, n_gauss_quad, max_gauss_pts, gauss_Ds, secants);
    RRX_CATCH("rrx_lw_secants_array")
}

int rrx_lw_secants_array_f32(int ncol, int ngpt, int n_gauss_quad, int max_gauss_pts, const float* gauss_Ds, float* secants, void* stream)
{
    RRX_TRY
    lw_secants_array_kernel<float><<<rrx::ceil_div(size_t(ncol)*ngpt*n_gauss_quad, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
            ncol, ngpt, n_gauss_quad, max_gauss_pts, gauss_Ds, secants);
    RRX_CATCH("rrx_lw_secants_array")
}

int rrx_lw_solver_noscat_f64(
        int ncol, int nlay, int ngpt, Bool top_at_1, int nmus,
        const double* secants, const double* weights,
        const double* tau, const double* lay_source, const double* lev_source,
        const double* sfc_emis, const double* sfc_src, const double* inc_flux,
        double* flux_up, double* flux_dn,
        Bool do_broadband, double* flux_up_loc, double* flux_dn_loc,
        Bool do_jacobians, const double* sfc_src_jac, double* flux_up_jac, void* stream)
{
    return lw_solver_noscat_impl<double>(ncol, nlay, ngpt, top_at_1, nmus, secants, weights, tau, lay_source, lev_source,
            sfc_emis, sfc_src, inc_flux, flux_up, flux_dn, do_broadband, flux_up_loc, flux_dn_loc,
            do_jacobians, sfc_src_jac, flux_up_jac, stream);
}

int rrx_lw_solver_noscat_f32(
        int ncol, int nlay, int ngpt, Bool top_at_1, int nmus,
        const float* secants, const float* weights,
        const float* tau, const float* lay_source, const float* lev_source,
        const float* sfc_emis, const float* sfc_src, const float* inc_flux,
        float* flux_up, float* flux_dn,
        Bool do_broadband, float* flux_up_loc, float* flux_dn_loc,
        Bool do_jacobians, const float* sfc_src_jac, float* flux_up_jac, void* stream)
{
    return lw_solver_noscat_impl<float>(ncol, nlay, ngpt, top_at_1, nmus, secants, weights, tau, lay_source, lev_source,
            sfc_emis, sfc_src, inc_flux, flux_up, flux_dn, do_broadband, flux_up_loc, flux_dn_loc,
            do_jacobians, sfc_src_jac, flux_up_jac, stream);
}

#define RRX_DEFINE_LW_FRACTIONS(F, SFX) \
int rrx_lw_solver_noscat_fractions##SFX( \
        int ncol, int nlay, int ngpt, Bool top_at_1, const F* secants, const F* weights, \
        const F* tau, const F* pfrac, const F* blay, const F* blev, const int* gpoint_bands, \
        const F* sfc_emis, const F* sfc_src, const F* inc_flux, F* flux_up_loc, F* flux_dn_loc, void* stream) \
{ \
    return lw_solver_noscat_fractions_impl<F>(ncol, nlay, ngpt, top_at_1, secants, weights, tau, pfrac, blay, blev, gpoint_bands, \
            sfc_emis, sfc_src, inc_flux, flux_up_loc, flux_dn_loc, stream); \
} \
int rrx_planck_sources_from_fractions##SFX(int ncol, int nlay, int ngpt, const int* gpoint_bands, const F* pfrac, const F* blay, \
        const F* blev, F* lay_src, F* lev_src, void* stream) \
{ return planck_sources_from_fractions_impl<F>(ncol, nlay, ngpt, gpoint_bands, pfrac, blay, blev, lay_src, lev_src, stream); }

RRX_DEFINE_LW_FRACTIONS(double, _f64)
RRX_DEFINE_LW_FRACTIONS(float, _f32)
}
