// Kernel laboratory for the fp64 fused-broadband LW no-scattering solver: variants of the production kernel
// (rte-rrtmgp-cpp_amd/csrc/rrx_solver_lw.hip) side by side in ONE process at the C4 shape, each checked against a plain
// serial kernel. Variants: lean exp, software-pipelined loads of the next g-point, and the "Planck-lite" form in which the
// solver rebuilds lay_source = pfrac*B_lay and lev_source = sqrt(pfrac*pfrac')*B_lev itself (reads 2 cell arrays instead of 3).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I rte-rrtmgp-cpp_amd/csrc -I include tools/lw_lab.hip -o tools/_build/lw_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>
#include "rrx_common.h"

#pragma clang fp contract(fast)
using namespace rrx;
typedef double F;

constexpr int GPB = 16;      // g-points per band in the lab

// ------------------------------------------------------------------------------------------------ reference
__global__ void serial_bb(const int ncol, const int nlay, const int ngpt, const F* __restrict__ tau, const F* __restrict__ lay,
        const F* __restrict__ lev, const F* __restrict__ emis, const F* __restrict__ ssrc, const F D, F* __restrict__ up, F* __restrict__ dn)
{
    const int icol = blockIdx.x*blockDim.x + threadIdx.x;
    if (icol >= ncol) return;
    const size_t ncl = ncol; const int nlev = nlay+1;
    const F pi = 3.14159265358979323846, tau_thres = sqrt(sqrt(DBL_EPSILON));
    for (int l=0; l<nlev; ++l) { up[l*ncl+icol] = 0; dn[l*ncl+icol] = 0; }
    for (int ig=0; ig<ngpt; ++ig)
    {
        const size_t lb = size_t(ig)*ncl*nlay + icol, vb = size_t(ig)*ncl*nlev + icol, sb = size_t(ig)*ncl + icol;
        auto layer = [&](int s, F& tr, F& sdn, F& sup)
        {
            const int ml = nlay-1-s;
            const F tl = tau[lb + ml*ncl]*D, ls = lay[lb + ml*ncl], la = lev[vb + (ml+1)*ncl], lbw = lev[vb + ml*ncl];
            tr = exp(-tl);
            const F fact = tl > tau_thres ? (1.-tr)/tl - tr : tl*(.5 + tl*(-1./3. + tl/8.));
            sdn = (1.-tr)*lbw + 2.*fact*(ls - lbw); sup = (1.-tr)*la + 2.*fact*(ls - la);
        };
        F d = 0.;
        for (int s=0; s<nlay; ++s) { F tr, a, b; layer(s, tr, a, b); dn[(nlay-s)*ncl+icol] += pi*d; d = tr*d + a; }
        dn[icol] += pi*d;
        F u = d*(1.-emis[sb]) + emis[sb]*ssrc[sb];
        up[icol] += pi*u;
        for (int s=nlay-1; s>=0; --s) { F tr, a, b; layer(s, tr, a, b); u = tr*u + b; up[(nlay-s)*ncl+icol] += pi*u; }
    }
}

// lay = pfrac*Blay, lev = sqrt(pfrac*pfrac_below)*Blev (memory order: layer 0 = surface), as planck_source_kernel does
__global__ void make_sources(const int ncol, const int nlay, const int ngpt, const F* __restrict__ pf, const F* __restrict__ blay,
        const F* __restrict__ blev, F* __restrict__ lay, F* __restrict__ lev)
{
    const size_t ncl = ncol; const int nlev = nlay+1;
    const size_t n = ncl*nlev*ngpt;
    for (size_t i = size_t(blockIdx.x)*blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x)*blockDim.x)
    {
        const int icol = int(i % ncl), m = int((i / ncl) % nlev), ig = int(i / (ncl*nlev));
        const int ib = ig / GPB;
        const size_t lb = size_t(ig)*ncl*nlay + icol;
        const F bl = blev[(size_t(ib)*nlev + m)*ncl + icol];
        F v;
        if (m == 0) v = pf[lb] * bl;
        else if (m == nlay) v = pf[lb + size_t(nlay-1)*ncl] * bl;
        else v = sqrt(pf[lb + size_t(m)*ncl] * pf[lb + size_t(m-1)*ncl]) * bl;
        lev[i] = v;
        if (m < nlay) lay[lb + size_t(m)*ncl] = pf[lb + size_t(m)*ncl] * blay[(size_t(ib)*nlay + m)*ncl + icol];
    }
}

// ------------------------------------------------------------------------------------------------ scan kernel
// FLAGS: 1 lean exp; 2 software-pipelined loads of the next g-point; 4 Planck-lite inputs (pfrac + per-band B arrays)
template<int K, int W, int CLT, int FLAGS, int MINW, int EV = 2>
__global__ void __launch_bounds__(256, MINW)
lw_bb(const int ncol, const int nlay, const int ngpt, const F D,
      const F* __restrict__ tau, const F* __restrict__ lay_source, const F* __restrict__ lev_source,
      const F* __restrict__ blay, const F* __restrict__ blev,
      const F* __restrict__ sfc_emis, const F* __restrict__ sfc_src, F* __restrict__ flux_up, F* __restrict__ flux_dn)
{
    constexpr int CL = CLT, LL = 64/CLT;
    constexpr bool PRE = (FLAGS & 2) != 0, LITE = (FLAGS & 4) != 0;
    constexpr int NA = LITE ? 2 : 3;           // cell arrays read per g-point
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & (CL-1), ll = lane / CL;
    const int h = wave % W, w0 = wave - h;
    const int wave_col0 = (blockIdx.x*(4/W) + wave/W) * CL;
    __shared__ F xch[4][4][CL];
    __shared__ F lds_b[LITE ? 2*K+1 : 1][256];        // per-thread columns: B_lay[K], B_lev[K+1] of the current band
    int icol = wave_col0 + cl;
    const bool active = icol < ncol;
    if (!active) icol = (wave_col0 < ncol) ? wave_col0 : 0;
    const bool writer = active && wave_col0 < ncol;
    const int nlev = nlay+1;
    const size_t ncl = size_t(ncol);
    const int t0 = (h*LL + ll)*K;
    const F pi = F(3.14159265358979323846);
    const F tau_thres = sqrt(sqrt(DBL_EPSILON));

    F acc_up[K], acc_dn[K];
    #pragma unroll
    for (int j=0; j<K; ++j) { acc_up[j] = F(0.); acc_dn[j] = F(0.); }

    // element offsets (top_at_1 = 0): layer s -> memory layer nlay-1-s; level t -> memory level nlay-t
    auto lay_off = [&](const int j) -> unsigned { return unsigned(max(nlay-1-t0-j, 0))*unsigned(ncol) + unsigned(icol); };
    auto lev_off = [&](const int j) -> unsigned { return unsigned(max(nlay-t0-j, 0))*unsigned(ncol) + unsigned(icol); };

    // loads of one g-point: a0 = tau, a1 = lay_source or pfrac, a2 = lev_source (levels t0..t0+K-1) [not LITE]
    F n0[PRE ? K : 1], n1[PRE ? K : 1], n2[(PRE && !LITE) ? K : 1], n_next = F(0.), n_prev = F(0.), n_emis = F(0.), n_ssrc = F(0.);
    auto issue = [&](const int g, F (&a0)[K], F (&a1)[K], F (&a2)[LITE ? 1 : K], F& x_next, F& x_prev, F& e, F& s)
    {
        const F* __restrict__ t_g = tau + size_t(g)*ncl*nlay;
        const F* __restrict__ l_g = lay_source + size_t(g)*ncl*nlay;
        #pragma unroll
        for (int j=0; j<K; ++j) { const unsigned o = lay_off(j); a0[j] = t_g[o]; a1[j] = l_g[o]; }
        if constexpr (!LITE)
        {
            const F* __restrict__ v_g = lev_source + size_t(g)*ncl*nlev;
            #pragma unroll
            for (int j=0; j<K; ++j) a2[j] = v_g[lev_off(j)];
            x_next = v_g[lev_off(K)];                 // level below the lane's last layer (clamped at the surface level)
        }
        else
        {
            x_next = l_g[lay_off(K)];                 // pfrac of the first layer of the next level-lane (clamped)
            x_prev = l_g[unsigned(min(max(nlay-1-(t0-1), 0), nlay-1))*unsigned(ncol) + unsigned(icol)];   // pfrac of layer t0-1 (clamped at the top)
        }
        e = sfc_emis[size_t(g)*ncl + icol]; s = sfc_src[size_t(g)*ncl + icol];
    };
    if constexpr (PRE) { F dummy[1]; if constexpr (LITE) issue(0, n0, n1, dummy, n_next, n_prev, n_emis, n_ssrc); else issue(0, n0, n1, n2, n_next, n_prev, n_emis, n_ssrc); }

    for (int igpt=0; igpt<ngpt; ++igpt)
    {
    if constexpr (!PRE) __syncthreads();
    F tv[K], ls[K], lvv[LITE ? 1 : K], x_next, x_prev = F(0.), emis, ssrc;
    if constexpr (PRE)
    {
        #pragma unroll
        for (int j=0; j<K; ++j) { tv[j] = n0[j]; ls[j] = n1[j]; if constexpr (!LITE) lvv[j] = n2[j]; }
        x_next = n_next; x_prev = n_prev; emis = n_emis; ssrc = n_ssrc;
    }
    else issue(igpt, tv, ls, lvv, x_next, x_prev, emis, ssrc);

    if constexpr (LITE)
    {
        if (igpt % GPB == 0)                  // band change: this lane's Planck functions into its LDS columns
        {
            const int ib = igpt / GPB;
            const F* __restrict__ bl = blay + size_t(ib)*ncl*nlay;
            const F* __restrict__ bv = blev + size_t(ib)*ncl*nlev;
            #pragma unroll
            for (int j=0; j<K; ++j) lds_b[j][tid] = bl[lay_off(j)];
            #pragma unroll
            for (int j=0; j<=K; ++j) lds_b[K+j][tid] = bv[lev_off(j)];
        }
    }

    // level source at level t0+j (LITE: sits between layers t0+j-1 and t0+j; the domain's first and last level use their
    // only neighbour). Evaluated inside the layer loop so that only two level values are live at a time.
    auto level_src = [&](const int j) -> F
    {
        if constexpr (!LITE) return (j < K) ? lvv[min(j, K-1)] : x_next;
        else
        {
            const int t = t0 + j;
            const F pa = (j == 0) ? x_prev : ls[max(j-1, 0)];
            const F pb = (j == K) ? x_next : ls[min(j, K-1)];
            const F bvv = lds_b[K+j][tid];
            if (t <= 0) return pb * bvv;
            if (t >= nlay) return pa * bvv;
            return ((FLAGS & 1) ? sqrt_pos(pa*pb) : sqrt(pa*pb)) * bvv;
        }
    };

    F tr[K], sdn[K], sup[K];
    F A = F(1.), Bdn = F(0.), Bup = F(0.);
    F lva = level_src(0);
    #pragma unroll
    for (int j=0; j<K; ++j)
    {
        const bool valid = (t0 + j) < nlay;
        F tvj = tv[j];
        if (j >= EV) asm volatile("" : "+v"(tvj) : "v"(sup[j-EV]));     // at most EV layer evaluations in flight
        const F lvb = level_src(j+1);
        const F lsj = LITE ? ls[j] * lds_b[j][tid] : ls[j];
        const F tau_loc = tvj * D;
        const F trans = (FLAGS & 1) ? exp_neg(-tau_loc) : exp(-tau_loc);
        const F fact = tau_loc > tau_thres ? (F(1.) - trans) * fast_rcp(tau_loc) - trans
                                           : tau_loc * (F(.5) + tau_loc * (F(-1./3.) + tau_loc * F(1./8.)));
        const F omt = F(1.) - trans;
        const F s_dn = omt * lvb + F(2.) * fact * (lsj - lvb);
        const F s_up = omt * lva + F(2.) * fact * (lsj - lva);
        lva = lvb;
        tr[j] = valid ? trans : F(1.); sdn[j] = valid ? s_dn : F(0.); sup[j] = valid ? s_up : F(0.);
        Bdn = tr[j]*Bdn + sdn[j];
        Bup += A*sup[j];
        A *= tr[j];
    }

    F dn_in, up_in;
    {
        F a = A, b = Bdn;
        #pragma unroll
        for (int d=1; d<LL; d<<=1)
        {
            const F a2 = shfl(a, lane - d*CL), b2 = shfl(b, lane - d*CL);
            if (ll >= d) { b = a*b2 + b; a = a*a2; }
        }
        F xa = F(1.), xb = F(0.), fa, fb;
        if (ll == LL-1) { xch[0][wave][cl] = a; xch[1][wave][cl] = b; }
        __syncthreads();
        if constexpr (PRE)
        {
            __builtin_amdgcn_sched_barrier(0);
            const int gn = min(igpt+1, ngpt-1);
            F dummy[1];
            if constexpr (LITE) issue(gn, n0, n1, dummy, n_next, n_prev, n_emis, n_ssrc); else issue(gn, n0, n1, n2, n_next, n_prev, n_emis, n_ssrc);
            __builtin_amdgcn_sched_barrier(0);
        }
        fa = F(1.); fb = F(0.);
        #pragma unroll
        for (int w=0; w<W; ++w)
        {
            const F oa = xch[0][w0+w][cl], ob = xch[1][w0+w][cl];
            if (w == h) { xa = fa; xb = fb; }
            fb = oa*fb + ob; fa = oa*fa;
        }
        if (h > 0) { b = a*xb + b; a = a*xa; }
        F be = shfl(b, lane - CL);
        if (ll == 0) be = xb;
        dn_in = be;
        const F dn_sfc = fb;
        const F up_sfc = dn_sfc * (F(1.) - emis) + emis * ssrc;

        a = A; b = Bup;
        #pragma unroll
        for (int d=1; d<LL; d<<=1)
        {
            const F a2 = shfl(a, lane + d*CL), b2 = shfl(b, lane + d*CL);
            if (ll + d < LL) { b = a*b2 + b; a = a*a2; }
        }
        xa = F(1.); xb = F(0.);
        if (ll == 0) { xch[2][wave][cl] = a; xch[3][wave][cl] = b; }
        __syncthreads();
        #pragma unroll
        for (int w=W-1; w>=1; --w)
            if (w > h) { const F oa = xch[2][w0+w][cl], ob = xch[3][w0+w][cl]; xb = oa*xb + ob; xa = oa*xa; }
        if (h < W-1) { b = a*xb + b; a = a*xa; }
        F ae = shfl(a, lane + CL); be = shfl(b, lane + CL);
        if (ll == LL-1) { ae = xa; be = xb; }
        up_in = ae*up_sfc + be;
    }
    {
        F dn = dn_in;
        #pragma unroll
        for (int j=0; j<K; ++j) { add_rounded(acc_dn[j], pi*dn); dn = tr[j]*dn + sdn[j]; }
        F up = up_in;
        #pragma unroll
        for (int j=K-1; j>=0; --j) { up = tr[j]*up + sup[j]; add_rounded(acc_up[j], pi*up); }
    }
    }
    if (!writer) return;
    #pragma unroll
    for (int j=0; j<K; ++j)
    {
        const int t = t0 + j;
        if (t <= nlay) { const size_t o = size_t(icol) + size_t(nlay - t)*ncl; flux_up[o] = acc_up[j]; flux_dn[o] = acc_dn[j]; }
    }
}

// ------------------------------------------------------------------------------------------------ host
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
struct Args { int ncol, nlay, ngpt; F D; const F *tau, *lay, *lev, *pf, *blay, *blev, *emis, *ssrc; F *up, *dn; };
struct Variant { std::string name; void (*launch)(const Args&); };

template<int K, int W, int CLT, int FLAGS, int MINW, int EV = 2>
void launch(const Args& a)
{
    constexpr int LL = 64/CLT;
    if ((a.nlay + 1 + LL*W - 1)/(LL*W) > K) { printf("K too small\n"); exit(1); }
    const dim3 grid((a.ncol + (4/W)*CLT - 1)/((4/W)*CLT));
    lw_bb<K,W,CLT,FLAGS,MINW,EV><<<grid, 256>>>(a.ncol, a.nlay, a.ngpt, a.D, a.tau, (FLAGS & 4) ? a.pf : a.lay, a.lev, a.blay, a.blev, a.emis, a.ssrc, a.up, a.dn);
}

int main(int argc, char** argv)
{
    const int ncol = argc > 1 ? atoi(argv[1]) : 16384;
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    const int nlay = 140, ngpt = 256, nlev = nlay+1, nbnd = ngpt/GPB;
    const int ncheck = std::min(ncol, 1024);
    const size_t ncell = size_t(ncol)*nlay*ngpt, nlv = size_t(ncol)*nlev*ngpt;
    std::vector<F> h_tau(ncell), h_pf(ncell), h_blay(size_t(ncol)*nlay*nbnd), h_blev(size_t(ncol)*nlev*nbnd), h_e(size_t(ncol)*ngpt), h_s(size_t(ncol)*ngpt);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return double(st >> 11) / 9007199254740992.0; };
    for (size_t i=0; i<ncell; ++i) { h_tau[i] = pow(10.0, -4 + 5*rnd()); h_pf[i] = 0.01 + 0.1*rnd(); }
    for (auto& x : h_blay) x = 50 + 300*rnd();
    for (auto& x : h_blev) x = 50 + 300*rnd();
    for (size_t i=0; i<h_e.size(); ++i) { h_e[i] = 0.8 + 0.2*rnd(); h_s[i] = 5 + 30*rnd(); }
    F *tau, *pf, *lay, *lev, *blay, *blev, *e, *s, *up, *dn, *rup, *rdn;
    CHECK(hipMalloc(&tau, ncell*8)); CHECK(hipMalloc(&pf, ncell*8)); CHECK(hipMalloc(&lay, ncell*8)); CHECK(hipMalloc(&lev, nlv*8));
    CHECK(hipMalloc(&blay, h_blay.size()*8)); CHECK(hipMalloc(&blev, h_blev.size()*8)); CHECK(hipMalloc(&e, h_e.size()*8)); CHECK(hipMalloc(&s, h_s.size()*8));
    const size_t nl = size_t(ncol)*nlev;
    CHECK(hipMalloc(&up, nl*8)); CHECK(hipMalloc(&dn, nl*8)); CHECK(hipMalloc(&rup, nl*8)); CHECK(hipMalloc(&rdn, nl*8));
    CHECK(hipMemcpy(tau, h_tau.data(), ncell*8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(pf, h_pf.data(), ncell*8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(blay, h_blay.data(), h_blay.size()*8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(blev, h_blev.data(), h_blev.size()*8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(e, h_e.data(), h_e.size()*8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(s, h_s.data(), h_s.size()*8, hipMemcpyHostToDevice));
    make_sources<<<2048, 256>>>(ncol, nlay, ngpt, pf, blay, blev, lay, lev);
    const F D = 1./0.6096748751;
    serial_bb<<<(ncheck+63)/64, 64>>>(ncol, nlay, ngpt, tau, lay, lev, e, s, D, rup, rdn);
    CHECK(hipDeviceSynchronize());
    std::vector<F> r_up(nl), r_dn(nl), g_up(nl), g_dn(nl);
    CHECK(hipMemcpy(r_up.data(), rup, nl*8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r_dn.data(), rdn, nl*8, hipMemcpyDeviceToHost));
    const Args a{ncol, nlay, ngpt, D, tau, lay, lev, pf, blay, blev, e, s, up, dn};

    std::vector<Variant> vs = {
        {"base   16x4 W4 K9                  ", launch<9,4,16,0,2>},
        {"exp    16x4 W4 K9                  ", launch<9,4,16,1,2>},
        {"exp+PRE 16x4 W4 K9                 ", launch<9,4,16,3,2>},
        {"exp+PRE 8x8 W2 K9                  ", launch<9,2,8,3,2>},
        {"exp     8x8 W2 K9                  ", launch<9,2,8,1,2>},
        {"LITE exp     16x4 W4 K9            ", launch<9,4,16,5,2>},
        {"LITE exp+PRE 16x4 W4 K9            ", launch<9,4,16,7,2>},
        {"LITE exp+PRE 8x8 W2 K9             ", launch<9,2,8,7,2>},
        {"LITE exp+PRE 8x8 W4 K5 minw3       ", launch<5,4,8,7,3>},
        {"exp+PRE 8x8 W4 K5 minw3            ", launch<5,4,8,3,3>},
        {"exp+PRE 16x4 W4 K9 EV1             ", launch<9,4,16,3,2,1>},
        {"exp+PRE 16x4 W4 K9 EV3             ", launch<9,4,16,3,2,3>},
        {"LITE exp+PRE 16x4 W4 K9 EV1        ", launch<9,4,16,7,2,1>},
        {"LITE exp+PRE 16x4 W4 K9 EV3        ", launch<9,4,16,7,2,3>},
    };
    const char* only = getenv("LAB_ONLY");
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<std::vector<float>> times(vs.size());
    for (int r=0; r<rounds+1; ++r)
        for (size_t v=0; v<vs.size(); ++v)
        {
            if (only && !strstr(vs[v].name.c_str(), only)) continue;
            CHECK(hipEventRecord(e0)); vs[v].launch(a); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) times[v].push_back(ms);
            if (r == 0)
            {
                CHECK(hipMemcpy(g_up.data(), up, nl*8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(g_dn.data(), dn, nl*8, hipMemcpyDeviceToHost));
                double worst = 0;
                for (int l=0; l<nlev; ++l) for (int c=0; c<ncheck; ++c)
                {
                    const size_t i = size_t(l)*ncol + c;
                    worst = std::max(worst, std::abs(g_up[i]-r_up[i]) / (std::abs(r_up[i]) + 1e-30));
                    worst = std::max(worst, std::abs(g_dn[i]-r_dn[i]) / (std::abs(r_dn[i]) + 1e-3));
                }
                printf("%s  max rel diff vs serial reference %.3e\n", vs[v].name.c_str(), worst);
            }
        }
    printf("\nncol %d x %d layers x %d g-points, %d rounds (median / min ms)\n", ncol, nlay, ngpt, rounds);
    for (size_t v=0; v<vs.size(); ++v)
    {
        if (times[v].empty()) continue;
        std::sort(times[v].begin(), times[v].end());
        printf("%s  %7.3f  %7.3f\n", vs[v].name.c_str(), times[v][times[v].size()/2], times[v][0]);
    }
    return 0;
}
