// EXPERIMENT RECORD (round 3) -- NOT built, NOT part of the product. The kernel and launcher text below were developed inside
// rte-rrtmgp-cpp_amd/csrc/rrx_solver_sw.hip (they use its two_stream<>, TwoStream, CL, LL, shfl, fast_rcp, add_rounded, sum_gpt_kernel)
// and measured on MI355X at C4 (16 384 columns x 140 layers x 256 g-points, fp64, clear sky) against the shipped
// sw_2stream_scan_kernel<double,1,9,2,BB,GZ,PRE> (4.68-4.85 ms, 256 VGPRs, two waves per SIMD):
//
//   geometry (waves per 8-column group x layers per lane, workgroup)      SW solver     note
//   W=3 K=6, 192 threads, 4 workgroups / CU, ds_bpermute scans             6.18 ms      WAIT_ANY 42-51 % of wave cycles
//   W=3 K=6, 384 threads (two column groups)                               8.26 ms
//   W=3 K=6, 768 threads (four column groups, one workgroup / CU)          5.85 ms
//   W=4 K=5, 256 threads, 3 workgroups / CU                                6.21 ms
//   W=6 K=3, 384 threads                                                   9.16 ms
//   W=3 K=6 with the all-to-all LDS exchanges below (this text)            7.18 ms      (and a defect left in its with-g path)
//
// Why three waves per SIMD do not pay here although they are what the fp64 pipe needs (profiles/r03_fp64_issue_costs.txt: 9 / 5.5 /
// 3.7 cycles per instruction with 1 / 2 / 3 waves): in-kernel s_memtime stamps (RRX_BB3_STAMPS) show the two-stream phase at the
// single-wave issue rate (K=6: 6.0 k cycles for 660 instructions) and everything else -- scans, replays, accumulators, 390
// instructions -- taking 14 k cycles per g-point: LDS round trips, four block barriers with 0.3-1.5 k cycles of skew each, and,
// decisive, s_waitcnt vmcnt(0) on scratch reloads of spilled loop invariants: vmcnt retires in order, so every such reload
// behind the software prefetch waits for the HBM loads of the next g-point. At 168 VGPRs the compiler spills 12-70 registers in
// every variant tried (EVALS, NPRE, prefetch point, buffer loads with scalar descriptors, identity-row exchanges without selects).
// The 256-VGPR kernel has no such reloads in its loop and runs at the two-wave issue ceiling (1 655 M instructions x 5.5 cycles =
// 4.4 ms); it stays the product kernel. See DESIGN.md section 4.2.

// ---------------------------------------------------------------------------------------------------------------------
// Fused broadband form, third generation: THREE resident waves per SIMD.
// Measured on MI355X (tools/fp64_issue_bench.hip, profiles/r03_fp64_issue_costs.txt): one wave issues an fp64 instruction every
// 9 cycles, two waves on a SIMD 5.5 cycles per instruction, and only three reach the 3.7 cycles of the pipe -- dependent or
// independent alike, so instruction-level parallelism buys nothing and the 256-VGPR kernel above (two waves per SIMD) runs at
// its issue ceiling (1 655 M wave-instructions x 5.5 cycles = its 4.7 ms). This form trades layers per lane for waves:
// W = 3 waves x 8 level-lanes x K <= 6 layers per column group of 8 columns, a workgroup of 192 threads, four of them per CU
// (<= 168 VGPRs, 38 KB LDS each). Same algebra and scan structure as sw_2stream_scan_kernel<.., BB, GZ, PRE>; differences:
//  * the cross-wave step of each scan composes the totals of up to W-1 other waves (prefix scans: the waves above, suffix
//    scans: the waves below), still one block barrier per scan -- and a barrier now stalls 3 waves, not 4;
//  * padding layers (level slots beyond the surface) are made transparent by zeroing their optical depth where it is loaded
//    (tau = 0 gives r_dif = 0, t_dif = 1 to an ulp, t_noscat = 1, r_dir = t_dir = the eps floor) instead of five selects on
//    the results of every layer: 8 VALU instructions per layer less;
//  * cross-lane moves stay ds_bpermute: they cost no VALU issue slot (a DPP / v_permlane form would cost 2-4 per value), and
//    the LDS pipe is far from full.
#ifdef RRX_BB3_STAMPS
// diagnostic build only: cycles (s_memtime) per phase of the g-point loop, summed over all waves (no output depends on them)
__device__ unsigned long long bb3_stamps[16];
#define BB3_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; }
#else
#define BB3_STAMP(i)
#endif
#ifndef RRX_BB3_EVALS
#define RRX_BB3_EVALS 2
#endif
#ifndef RRX_BB3_NPRE
#define RRX_BB3_NPRE 3
#endif
#ifndef RRX_BB3_XSPLIT
#define RRX_BB3_XSPLIT 4
#endif
#ifndef RRX_BB3_RBATCH
#define RRX_BB3_RBATCH 3
#endif
// Register / LDS plan (fp64, K = 6): the per-layer state r, t (-> p, alpha), source_up, source_dn (-> src, b) and the direct-beam
// sums stay in registers (5K doubles); albedo and the up / down sums live in per-thread LDS columns (3K doubles); of the next
// g-point only the first NPRE layers (and the three surface / top values) are loaded ahead, behind the first barrier -- the loads
// of the other layers go out at the top of the two-stream phase and land while the first NPRE layers are evaluated.
// Latency plan. In-kernel stamps (RRX_BB3_STAMPS) showed the dense two-stream phase running at the single-wave issue rate
// (9 cycles per fp64 instruction) and everything else at 36 cycles per instruction: LDS round trips of ~300 cycles under load,
// four per scan with ds_bpermute. So the level scans go through LDS "all to all" instead: every lane stores its composite, and
// after ONE round trip reads the (at most 7) composites of the level-lanes above / below it -- a lane that needs no more reads an
// identity slot, chosen by a select on the address -- and composes them in registers. The direct-beam product and the albedo
// Moebius composites (un-normalised 2x2 products, normalised once) share one round trip and one block barrier.
template<typename F, int K, int W, int G, bool GZ, bool GS>
__global__ void __launch_bounds__(64*W*G, 3)
sw_bb3_kernel(
        const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* __restrict__ tau, const F* __restrict__ ssa, const F* __restrict__ g, const F* __restrict__ mu0,
        const F* __restrict__ sfc_alb_dir, const F* __restrict__ sfc_alb_dif,
        const F* __restrict__ inc_flux_dir, const F* __restrict__ inc_flux_dif,
        F* __restrict__ flux_up, F* __restrict__ flux_dn, F* __restrict__ flux_dir, const int gper)
{
    constexpr int T = 64*W*G;            // G column groups (8 columns each) per workgroup, W waves per group
    constexpr int NP = (RRX_BB3_NPRE < K) ? RRX_BB3_NPRE : K;      // layers of the next g-point loaded ahead
    typedef F Pair __attribute__((ext_vector_type(2)));
    __shared__ F lds_alb[K][T];      // albedo at the lane's levels
    __shared__ F lds_up[K][T];
    __shared__ F lds_dn[K][T];       // diffuse part; the direct beam is added at the end
    // In-wave exchange: per wave and component 8 data rows (one per level-lane) followed by 7 identity rows (written once). A lane
    // at row r reads rows r+1 .. r+7 -- ONE base address, immediate offsets, no selects: what lies beyond the wave is the identity.
    // Scans towards the surface store at row ll, scans towards the top at row 7-ll. Identities: components 0, 1 -> 1; 2, 3 -> 0.
    __shared__ F xa[W*G][4][2*LL-1][CL];
    __shared__ F xch[8][W*G][CL];    // wave totals (cross-wave step of each scan)
    (void)sizeof(Pair);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int bwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w0 = (bwave / W) * W;          // first wave of this column group inside the workgroup
    const int wave = bwave - w0;             // position of this wave in its column group (0 = top of the atmosphere)
    const int cl = lane & (CL-1);
    const int ll = lane >> 3;
    int icol = (blockIdx.x*G + bwave / W)*CL + cl;
    const bool writer = icol < ncol;
    if (!writer) icol = ncol - 1;                       // lanes without a column shadow a valid one (no stores)
    const int nlev = nlay + 1;
    const size_t ncl = size_t(ncol);
    const int t0 = (wave*LL + ll)*K;

    const F mu = mu0[icol];
    const F mu_inv = F(1.)/mu;

    F acc_dir[K];
    #pragma unroll
    for (int j=0; j<K; ++j) { acc_dir[j] = F(0.); lds_up[j][tid] = F(0.); lds_dn[j][tid] = F(0.); }
    if (ll < LL-1)
    {
        xa[bwave][0][LL+ll][cl] = F(1.); xa[bwave][1][LL+ll][cl] = F(1.); xa[bwave][2][LL+ll][cl] = F(0.); xa[bwave][3][LL+ll][cl] = F(0.);
    }
    F* const x_dn = &xa[bwave][0][ll][cl];          // this lane's row in scans towards the surface (component 0; others at +q*XQ)
    F* const x_up = &xa[bwave][0][LL-1-ll][cl];     // ... in scans towards the top of the atmosphere
    constexpr int XQ = (2*LL-1)*CL;                 // words between the components of the exchange

    const int g_begin = GS ? blockIdx.y*gper : 0;
    const int g_end = GS ? min(ngpt, g_begin + gper) : ngpt;

    // Buffer loads: a descriptor per g-point slab in scalar registers + a 32-bit byte offset per lane and layer (the launcher
    // checks that a slab stays below 4 GB) -- no 64-bit lane address is ever kept in vector registers
    auto slab_of = [](const F* base, const unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<F*>(base), 0, int(bytes), 0x00020000); };
    auto at = [](const __amdgpu_buffer_rsrc_t r, const unsigned byte_off) -> F
    {
        if constexpr (sizeof(F) == 8) return __builtin_bit_cast(F, __builtin_amdgcn_raw_buffer_load_b64(r, int(byte_off), 0, 0));
        else return __builtin_bit_cast(F, __builtin_amdgcn_raw_buffer_load_b32(r, int(byte_off), 0, 0));
    };
    auto off_of = [&](const int j) -> unsigned
    {
        const int ml = top_at_1 ? min(t0 + j, nlay-1) : max(nlay-1-t0-j, 0);
        return (unsigned(ml)*unsigned(ncol) + unsigned(icol)) * unsigned(sizeof(F));
    };
    const unsigned col_off = unsigned(icol) * unsigned(sizeof(F));
    const unsigned slab_bytes = unsigned(ncol)*unsigned(nlay)*unsigned(sizeof(F)), row_bytes = unsigned(ncol)*unsigned(sizeof(F));
    // one layer of one g-point; tau of a padding layer (level slot beyond the surface) is zeroed: the layer becomes transparent
    auto load_layer = [&](const int ig, const int j, F& tv, F& wv, F& gv)
    {
        const size_t slab = size_t(ig)*ncl*nlay;
        const unsigned o = off_of(j);
        const F t_ = at(slab_of(tau + slab, slab_bytes), o);
        tv = (t0 + j < nlay) ? t_ : F(0.);
        wv = at(slab_of(ssa + slab, slab_bytes), o);
        if constexpr (!GZ) gv = at(slab_of(g + slab, slab_bytes), o); else gv = F(0.);
    };
    F pt[NP], pw[NP], pg[GZ ? 1 : NP], n_inc, n_adir, n_adif;
    auto load_ahead = [&](const int ig)
    {
        #pragma unroll
        for (int j=0; j<NP; ++j) { F gv; load_layer(ig, j, pt[j], pw[j], gv); if constexpr (!GZ) pg[j] = gv; }
        const size_t s0 = size_t(ig)*ncl;
        n_inc = at(slab_of(inc_flux_dir + s0, row_bytes), col_off); n_adir = at(slab_of(sfc_alb_dir + s0, row_bytes), col_off);
        n_adif = at(slab_of(sfc_alb_dif + s0, row_bytes), col_off);
    };
    load_ahead(g_begin);
#ifdef RRX_BB3_STAMPS
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif

    for (int igpt=g_begin; igpt<g_end; ++igpt)
    {
        F rp[K], al[K], sb[K], qb[K];      // names follow their LAST meaning (see sw_2stream_scan_kernel)
        F tp[K];                           // transmittance of the direct beam from the top of the lane's chunk to layer j

        // ---- (a) two-stream coefficients and the direct beam relative to the lane's incoming beam
        F ct[K > NP ? K-NP : 1], cw[K > NP ? K-NP : 1], cg[(!GZ && K > NP) ? K-NP : 1];
        #pragma unroll
        for (int j=NP; j<K; ++j) { F gv; load_layer(igpt, j, ct[j-NP], cw[j-NP], gv); if constexpr (!GZ) cg[j-NP] = gv; }
        __builtin_amdgcn_sched_barrier(0);
        F Tloc = F(1.);
        #pragma unroll
        for (int j=0; j<K; ++j)
        {
            F tv = (j < NP) ? pt[j < NP ? j : 0] : ct[j >= NP ? j-NP : 0];
            const F wv = (j < NP) ? pw[j < NP ? j : 0] : cw[j >= NP ? j-NP : 0];
            F gv = F(0.);
            if constexpr (!GZ) gv = (j < NP) ? pg[j < NP ? j : 0] : cg[j >= NP ? j-NP : 0];
            if (j >= RRX_BB3_EVALS) asm volatile("" : "+v"(tv) : "v"(qb[j-RRX_BB3_EVALS]));      // at most that many evaluations in flight
            const TwoStream<F> ts = two_stream<F,GZ>(tv, wv, gv, mu, mu_inv);
            rp[j] = ts.r_dif; al[j] = ts.t_dif;
            sb[j] = ts.r_dir * Tloc; qb[j] = ts.t_dir * Tloc;
            tp[j] = Tloc;
            Tloc *= ts.t_noscat;
        }
        __builtin_amdgcn_sched_barrier(0);
        BB3_STAMP(0)
        const F inc_dir = n_inc, a_dir = n_adir, a_dif = n_adif;
        F inc_dif = F(0.);
        if (inc_flux_dif != nullptr) inc_dif = at(slab_of(inc_flux_dif + size_t(igpt)*ncl, row_bytes), col_off);

        // ---- (b) Moebius composite of the lane's layers for the albedo recurrence (layer K-1 applied first), normalised to m11 = 1
        F m00 = F(1.), m01 = F(0.), m10 = F(0.), m11 = F(1.);
        #pragma unroll
        for (int j=K-1; j>=0; --j)
        {
            const F r = rp[j], t = al[j];
            const F e = t*t - r*r;
            const F n00 = e*m00 + r*m10, n01 = e*m01 + r*m11;
            const F n10 = m10 - r*m00,   n11 = m11 - r*m01;
            m00 = n00; m01 = n01; m10 = n10; m11 = n11;
        }
        {
            const F inv = fast_rcp(m11);
            m00 *= inv; m01 *= inv; m10 *= inv;
        }
        // ---- (c) merged in-wave exchange: direct-beam product of the level-lanes above, albedo composite of the level-lanes below
        x_up[0] = Tloc;
        x_dn[XQ] = m00; x_dn[2*XQ] = m01; x_dn[3*XQ] = m10;
        F pe = F(1.);                                           // product of the level-lanes above (exclusive)
        #pragma unroll
        for (int k=1; k<LL; ++k) pe *= x_up[k*CL];
        __builtin_amdgcn_sched_barrier(0);
        F e00 = F(1.), e01 = F(0.), e10 = F(0.), e11 = F(1.);   // composite of the level-lanes below (exclusive), un-normalised
        #pragma unroll
        for (int k=1; k<LL; ++k)
        {
            const F y00 = x_dn[XQ + k*CL], y01 = x_dn[2*XQ + k*CL], y10 = x_dn[3*XQ + k*CL];      // E = E * Y (Y normalised: y11 = 1)
            const F n00 = e00*y00 + e01*y10, n01 = e00*y01 + e01;
            const F n10 = e10*y00 + e11*y10, n11 = e10*y01 + e11;
            e00 = n00; e01 = n01; e10 = n10; e11 = n11;
        }
        {
            // wave totals: product down to the bottom of level-lane 7, composite from the top of level-lane 0 (normalised)
            const F s00 = m00*e00 + m01*e10, s01 = m00*e01 + m01*e11, s10 = m10*e00 + e10, s11 = m10*e01 + e11;
            const F inv = fast_rcp(s11);
            if (ll == LL-1) xch[0][bwave][cl] = pe * Tloc;
            if (ll == 0) { xch[1][bwave][cl] = s00*inv; xch[2][bwave][cl] = s01*inv; xch[3][bwave][cl] = s10*inv; }
        }
        BB3_STAMP(1)
        __syncthreads();
        BB3_STAMP(2)
        F ptot = F(1.);
        #pragma unroll
        for (int u=0; u<W; ++u)
        {
            const F o = xch[0][w0+u][cl];
            if (u < wave) pe *= o;
            ptot *= o;
        }
        const F dir_top = inc_dir * mu;
        const F dir_in = dir_top * pe;
        const F dir_sfc = dir_top * ptot;
        #pragma unroll
        for (int j=0; j<K; ++j)
        {
            sb[j] *= dir_in; qb[j] *= dir_in;
            add_rounded(acc_dir[j], dir_in * tp[j]);
        }
        // every wave of the workgroup has passed the barrier: the first loads of the next g-point go out together and land during
        // the scans (the prefix transmittances above are dead, their registers take the loads)
        __builtin_amdgcn_sched_barrier(0);
        load_ahead(min(igpt + 1, g_end - 1));                       // (last iteration: a harmless re-read)
        __builtin_amdgcn_sched_barrier(0);

        // albedo at the bottom of this wave's levels: the surface albedo through the composites of the waves below
        F awb = a_dif;
        #pragma unroll
        for (int u=W-1; u>=1; --u)
            if (u > wave) awb = (xch[1][w0+u][cl]*awb + xch[2][w0+u][cl]) * fast_rcp(xch[3][w0+u][cl]*awb + F(1.));
        F a = (e00*awb + e01) * fast_rcp(e10*awb + e11);            // albedo at the bottom of this lane's chunk

        // replay albedo upward; alpha, beta, p, q and the lane's affine composites
        F As = F(1.), Bs = F(0.), Bd = F(0.);
        #pragma unroll
        for (int j=K-1; j>=0; --j)
        {
            const F r = rp[j], t = al[j];
            const F denom = fast_rcp(F(1.) - r*a);
            const F alpha = t*denom;
            const F beta = sb[j] + alpha*a*qb[j];
            a = r + t*alpha*a;
            lds_alb[j][tid] = a;
            al[j] = alpha;
            sb[j] = beta;
            rp[j] = r*denom;
            qb[j] = qb[j]*denom;
            Bs = alpha*Bs + beta;
            As *= alpha;
        }

        // ---- (d) source: affine composite of the level-lanes below (exclusive), then of the waves below
        x_dn[XQ] = As; x_dn[2*XQ] = Bs;
        F ea = F(1.), eb = F(0.);
        #pragma unroll
        for (int k=1; k<LL; ++k)
        {
            const F ya = x_dn[XQ + k*CL], yb = x_dn[2*XQ + k*CL];
            eb = ea*yb + eb; ea = ea*ya;                                     // E = E o T: the next level-lane down acts first
        }
        if (ll == 0) { xch[4][bwave][cl] = As*ea; xch[5][bwave][cl] = As*eb + Bs; }
        BB3_STAMP(3)
        __syncthreads();
        BB3_STAMP(4)
        F s = dir_sfc * a_dir;                                               // src at the surface
        #pragma unroll
        for (int u=W-1; u>=1; --u)
            if (u > wave) s = xch[4][w0+u][cl]*s + xch[5][w0+u][cl];          // -> at the bottom of this wave's levels
        s = ea*s + eb;                                                        // -> at the bottom of this lane's chunk
        F Q = F(1.);
        #pragma unroll
        for (int j=K-1; j>=0; --j)
        {
            const F b = rp[j]*s + qb[j];
            s = al[j]*s + sb[j];
            sb[j] = s;
            qb[j] = b;
            Bd += Q*b;
            Q *= al[j];
        }

        // ---- (e) diffuse down: affine composite of the level-lanes above (exclusive), then of the waves above
        x_up[XQ] = As; x_up[2*XQ] = Bd;
        ea = F(1.); eb = F(0.);
        #pragma unroll
        for (int k=1; k<LL; ++k)
        {
            const F ya = x_up[XQ + k*CL], yb = x_up[2*XQ + k*CL];
            eb = ea*yb + eb; ea = ea*ya;                                     // E = E o T: the next level-lane up acts first
        }
        if (ll == LL-1) { xch[6][bwave][cl] = As*ea; xch[7][bwave][cl] = As*eb + Bd; }
        BB3_STAMP(5)
        __syncthreads();
        BB3_STAMP(6)
        F dn = inc_dif;                                                      // diffuse flux at the top of the atmosphere
        #pragma unroll
        for (int u=0; u<W-1; ++u)
            if (u < wave) dn = xch[6][w0+u][cl]*dn + xch[7][w0+u][cl];       // -> at the top of this wave's levels
        dn = ea*dn + eb;                                                     // -> at the top of this lane's chunk

        // ---- replay the diffuse downward flux; add this g-point to the sums (g-point order, one rounded addition each)
        #pragma unroll
        for (int j0=0; j0<K; j0+=RRX_BB3_RBATCH)        // the LDS reads of RBATCH levels go out together
        {
            constexpr int NB = RRX_BB3_RBATCH;
            F alb[NB], au[NB], ad[NB];
            #pragma unroll
            for (int i=0; i<NB; ++i) if (j0 + i < K) { alb[i] = lds_alb[j0+i][tid]; au[i] = lds_up[j0+i][tid]; ad[i] = lds_dn[j0+i][tid]; }
            #pragma unroll
            for (int i=0; i<NB; ++i) if (j0 + i < K)
            {
                const int j = j0 + i;
                const F ou = dn*alb[i] + sb[j];
                add_rounded(au[i], ou); add_rounded(ad[i], dn);
                lds_up[j][tid] = au[i]; lds_dn[j][tid] = ad[i];
                dn = al[j]*dn + qb[j];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        BB3_STAMP(9)
    }   // g-point loop
#ifdef RRX_BB3_STAMPS
    if (lane == 0) for (int i=0; i<10; ++i) atomicAdd(&bb3_stamps[i], st_acc[i]);
#endif

    if (!writer) return;
    #pragma unroll
    for (int j=0; j<K; ++j)
    {
        const int t = t0 + j;
        if (t <= nlay)
        {
            const int ml = top_at_1 ? t : nlay - t;
            const size_t o = size_t(icol) + size_t(ml)*ncl + (GS ? size_t(blockIdx.y)*ncl*nlev : size_t(0));
            flux_up[o] = lds_up[j][tid]; flux_dn[o] = lds_dn[j][tid] + acc_dir[j]; flux_dir[o] = acc_dir[j];
        }
    }
}



// third-generation fused broadband form (three waves per SIMD): fp64; W waves per column group, G column groups per workgroup
template<typename F, int W, int G, int KMAX>
bool launch_bb3(hipStream_t st,
        const int ncol, const int nlay, const int ngpt, const int top_at_1,
        const F* tau, const F* ssa, const F* g, const F* mu0, const F* sfc_alb_dir, const F* sfc_alb_dif,
        const F* inc_flux_dir, const F* inc_flux_dif, F* flux_up, F* flux_dn, F* flux_dir)
{
    const int need = std::max(4, ceil_div(nlay+1, LL*W));
    if (need > KMAX || size_t(ncol)*nlay*sizeof(F) >= (size_t(1) << 32)) return false;        // (32-bit byte offsets inside a g-point slab)
    const int groups = ceil_div(ncol, CL*G);
    const int gper = ceil_div(ngpt, broadband_gsplit(ceil_div(groups*G, 2), ngpt));
    const int nsplit = ceil_div(ngpt, gper);
    const size_t nlevcol = size_t(ncol)*(nlay+1);
    StreamScratch scratch(st);
    F* up = flux_up; F* dn = flux_dn; F* dr = flux_dir;
    if (nsplit > 1) { up = scratch.get<F>(3*nsplit*nlevcol); dn = up + nsplit*nlevcol; dr = dn + nsplit*nlevcol; }
    const dim3 grid(groups, nsplit);
    if (std::getenv("RRX_OCC") != nullptr)
    {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sw_bb3_kernel<F,KMAX,W,G,true,false>, 64*W*G, 0);
        std::fprintf(stderr, "[sw_bb3] occupancy query: %d workgroups of %d threads per CU\n", nb, 64*W*G);
    }
#define RRX_SW_K3(KK) if (KK <= KMAX && need <= KK) { constexpr int KC = (KK <= KMAX) ? KK : KMAX; \
        if (nsplit > 1 && g == nullptr) sw_bb3_kernel<F,KC,W,G,true,true><<<grid, 64*W*G, 0, st>>>(ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, \
            sfc_alb_dir, sfc_alb_dif, inc_flux_dir, inc_flux_dif, up, dn, dr, gper); \
        else if (nsplit > 1) sw_bb3_kernel<F,KC,W,G,false,true><<<grid, 64*W*G, 0, st>>>(ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, \
            sfc_alb_dir, sfc_alb_dif, inc_flux_dir, inc_flux_dif, up, dn, dr, gper); \
        else if (g == nullptr) sw_bb3_kernel<F,KC,W,G,true,false><<<grid, 64*W*G, 0, st>>>(ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, \
            sfc_alb_dir, sfc_alb_dif, inc_flux_dir, inc_flux_dif, up, dn, dr, gper); \
        else sw_bb3_kernel<F,KC,W,G,false,false><<<grid, 64*W*G, 0, st>>>(ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, \
            sfc_alb_dir, sfc_alb_dif, inc_flux_dir, inc_flux_dif, up, dn, dr, gper); \
        break; }
#ifdef RRX_BB3_LAB
    do { RRX_SW_K3(KMAX) } while (false);               // laboratory builds: only the largest K of each geometry
#else
    do { RRX_SW_K3(4) RRX_SW_K3(5) RRX_SW_K3(6) } while (false);
#endif
#undef RRX_SW_K3
#ifdef RRX_BB3_STAMPS
    {
        unsigned long long h[16] = {};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(bb3_stamps), sizeof(h));
        const double per = 1.0 / (double(groups)*nsplit*W*G*double(gper));          // per wave and g-point
        static const char* names[10] = {"two-stream", "Moebius composite + merged exchange", "barrier 1", "dir acc + prefetch + albedo replay + src exchange", "barrier 2",
                                        "src replay + down exchange", "barrier 3", "-", "-", "final replay"};
        double tot = 0; for (int i=0; i<10; ++i) tot += double(h[i])*per;
        std::fprintf(stderr, "[sw_bb3 W=%d G=%d] cycles per wave and g-point: total %.0f\n", W, G, tot);
        for (int i=0; i<10; ++i) std::fprintf(stderr, "    %-40s %8.0f\n", names[i], double(h[i])*per);
        unsigned long long z[16] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(bb3_stamps), z, sizeof(z));
    }
#endif
    if (nsplit > 1)
    {
        const int nb = ceil_div(nlevcol, 256);
        sum_gpt_kernel<F><<<nb, 256, 0, st>>>(nlevcol, nsplit, up, flux_up);
        sum_gpt_kernel<F><<<nb, 256, 0, st>>>(nlevcol, nsplit, dn, flux_dn);
        sum_gpt_kernel<F><<<nb, 256, 0, st>>>(nlevcol, nsplit, dr, flux_dir);
    }
    return true;
}

