// Shared device/host helpers for the rrx HIP kernels (gfx950 / CDNA4 only).
#ifndef RRX_COMMON_H
#define RRX_COMMON_H

#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdlib>
#include <cmath>
#include <string>
#include <vector>
#include <stdexcept>

typedef signed char Bool;   // RTE_USE_CBOOL in every shipped reference config (config/ubuntu_22lts.cmake:35)

namespace rrx
{
    // ---- per-thread settings of the device layer (re-entrant per device / host thread). The rrx_set_*
    //      entry points change the CALLING thread's copy only; a thread that never calls them runs the defaults.
    struct Tuning
    {
        int lw_variant = 0;        // rrx_set_lw_variant: kernel tiling for A/B runs (0 = default)
        int sw_variant = 0;        // rrx_set_sw_variant
        int bb_min_groups = 512;   // rrx_set_broadband_min_groups: column groups needed for the one-pass broadband form
        int bb_gsplit = 0;         // rrx_set_broadband_gsplit: g-point ranges per column group in that form (0 = as many as it
                                   // takes to reach bb_min_groups workgroups, 1 = never split)
        int sync_waves = 1;        // partner waves issue their load bursts together (env RRX_SYNC, default on)
        int go_share = 1;          // Planck shared-cell path (env RRX_GO_SHARE, default on)
        int go_window = 1;         // windowed gas optics ahead of the gather kernel (env RRX_GO_WINDOW, default on)
    };
    Tuning& tuning();              // defined in rrx_misc.hip (thread_local)

    // ---- error plumbing: C-ABI functions return int, message retrievable with rrx_last_error() ----
    void set_error(const std::string& msg);
    int check_launch(const char* what);

    constexpr int WAVE = 64;

    template<typename F> struct Lim;
    template<> struct Lim<double>
    {
        __host__ __device__ static constexpr double eps() { return DBL_EPSILON; }
        __host__ __device__ static constexpr double tiny() { return DBL_MIN; }
        __host__ __device__ static constexpr double k_min() { return 1.e-12; }
    };
    template<> struct Lim<float>
    {
        __host__ __device__ static constexpr float eps() { return FLT_EPSILON; }
        __host__ __device__ static constexpr float tiny() { return FLT_MIN; }
        __host__ __device__ static constexpr float k_min() { return 1.e-4f; }
    };

    // ---- cross-lane moves (wave64). ds_bpermute-based shuffles of 32/64-bit values ----
    __device__ __forceinline__ float shfl(const float v, const int src_lane) { return __shfl(v, src_lane, WAVE); }
    __device__ __forceinline__ double shfl(const double v, const int src_lane) { return __shfl(v, src_lane, WAVE); }

    // ---- vector-of-columns helpers: V consecutive columns handled by one lane ----
#ifndef RRX_NT_LOADS
#define RRX_NT_LOADS 0    // A/B: non-temporal loads of the solvers' cell arrays
#endif
#ifndef RRX_NT_STORES
#define RRX_NT_STORES 0   // A/B: non-temporal stores of the solvers' flux arrays
#endif
    template<typename F, int V> struct Vec { F v[V]; };

    // V consecutive columns starting at p. The launchers only pick V > 1 when ncol % V == 0, so a lane's V columns
    // are either all inside the array or the lane is inactive (and then loads a valid dummy address).
    template<typename F, int V>
    __device__ __forceinline__ Vec<F,V> load_cols(const F* __restrict__ p)
    {
        Vec<F,V> r;
        if constexpr (V == 1)
        {
#if RRX_NT_LOADS
            r.v[0] = __builtin_nontemporal_load(p);
#else
            r.v[0] = p[0];
#endif
        }
        else
        {
            typedef F vecT __attribute__((ext_vector_type(V)));
#if RRX_NT_LOADS
            const vecT t = __builtin_nontemporal_load(reinterpret_cast<const vecT*>(p));
#else
            const vecT t = *reinterpret_cast<const vecT*>(p);
#endif
            #pragma unroll
            for (int i=0; i<V; ++i) r.v[i] = t[i];
        }
        return r;
    }

    template<typename F, int V>
    __device__ __forceinline__ void store_cols(F* __restrict__ p, const Vec<F,V>& r)
    {
        if constexpr (V == 1)
        {
#if RRX_NT_STORES
            __builtin_nontemporal_store(r.v[0], p);
#else
            p[0] = r.v[0];
#endif
        }
        else
        {
            typedef F vecT __attribute__((ext_vector_type(V)));
            vecT t;
            #pragma unroll
            for (int i=0; i<V; ++i) t[i] = r.v[i];
#if RRX_NT_STORES
            __builtin_nontemporal_store(t, reinterpret_cast<vecT*>(p));
#else
            *reinterpret_cast<vecT*>(p) = t;
#endif
        }
    }

    // acc += x as a separate rounded addition: never fused with the multiplication that produced x, whatever the
    // contraction mode of the calling file (the fused broadband sums must round like sum_broadband on stored fluxes)
    template<typename F>
    __device__ __forceinline__ void add_rounded(F& acc, const F x)
    {
        #pragma clang fp contract(off)
        acc = acc + x;
    }

    // 1/x to ~1 ulp without the IEEE corner-case handling of a full division (operands here are finite, normal and
    // well away from 0/inf: layer transmissivities, two-stream denominators): v_rcp + 2 Newton steps (f64), 1 (f32).
    __device__ __forceinline__ double fast_rcp(const double x)
    {
        double r = __builtin_amdgcn_rcp(x);
        double e = fma(-x, r, 1.0); r = fma(r, e, r);
        e = fma(-x, r, 1.0); r = fma(r, e, r);
        return r;
    }
#ifndef RRX_F32_IEEE_RCP
    __device__ __forceinline__ float fast_rcp(const float x)
    {
        float r = __builtin_amdgcn_rcpf(x);
        const float e = fmaf(-x, r, 1.0f); r = fmaf(r, e, r);
        return r;
    }
#else
    __device__ __forceinline__ float fast_rcp(const float x) { return 1.0f / x; }
#endif

    // exp(x) for finite x <= 0 (layer transmissivities exp(-tau*k), exp(-tau/mu0)): libm's exp without its overflow /
    // special-value handling. Range reduction x = n ln2 + r, |r| <= ln2/2 (ln2 split so that n*ln2_hi is exact), minimax
    // polynomial of degree 11 for exp(r) (fitted at Chebyshev nodes; approximation error 1.6e-17), 2^n through ldexp, which
    // also delivers the underflow to 0. Measured against glibc on 2e7
    // arguments in [-1e3, -1e-8]: at most 1.0 ulp. 19 instructions against about 27 for the library call.
    __device__ __forceinline__ double exp_neg(const double x)
    {
        // (no clamp of x: n fits an int for x > -1.4e9, far beyond any finite optical depth x secant; round 3 clamped at -1000 with a
        //  v_max_f64 per evaluation)
        const double n = __builtin_rint(x * 0x1.71547652b82fep+0);
        double r = fma(n, -0x1.62e42f0000000p-1, x);
        r = fma(n, -0x1.df473de6af279p-26, r);
        double p = 0x1.af389ecfc4b9cp-26;
        p = fma(p, r, 0x1.28917c89a43a7p-22); p = fma(p, r, 0x1.71de0db2f6b19p-19); p = fma(p, r, 0x1.a019b9149a41cp-16);
        p = fma(p, r, 0x1.a01a01a7c2efep-13); p = fma(p, r, 0x1.6c16c17889ef1p-10); p = fma(p, r, 0x1.11111111109b5p-7);
        p = fma(p, r, 0x1.5555555553d68p-5); p = fma(p, r, 0x1.5555555555556p-3); p = fma(p, r, 0x1.0000000000001p-1);
        const double t = fma(r*r, p, r);
        return __builtin_amdgcn_ldexp(t + 1.0, (int)n);
    }
    // The same function through a 64-entry table of 2^(j/64) in LDS (the fused broadband solvers, whose bound is fp64 issue):
    // x = (64 e + j) ln2/64 + r, |r| <= ln2/128, exp(x) = 2^e * T[j] * (1 + r + r^2 p(r)) with a degree-3 p (truncation 3.5e-17). The
    // integer 64 e + j is read from the low word of x*64/ln2 + 1.5*2^52 (no conversion instruction); e comes from a saturating
    // conversion so that an absurd argument still ends in ldexp(.., INT_MIN) = 0. 15 vector instructions + one ds_read against 19;
    // at most 1.3 ulp from glibc on 2e7 arguments in [-1e3, -1e-8] (host model of the same arithmetic).
    static __device__ const double exp2_64_table[64] = {
        0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0, 0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0,
        0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0, 0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
        0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0, 0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0,
        0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0, 0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
        0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0, 0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0,
        0x1.6247eb03a5585p+0, 0x1.6623882552225p+0, 0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
        0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0, 0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0,
        0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0, 0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
        0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0, 0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0,
        0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0, 0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
        0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};
    __device__ __forceinline__ void exp_table_fill(double* lds_table)      // (the caller places a barrier before the first use)
    {
        if (threadIdx.x < 64) lds_table[threadIdx.x] = exp2_64_table[threadIdx.x];
    }
    __device__ __forceinline__ double exp_neg(const double x, const double* lds_table)
    {
        const double magic = 0x1.8p+52;
        const double t = fma(x, 0x1.71547652b82fep+6, magic);
        const double nf = t - magic;
        double r = fma(nf, -0x1.62e42fefa0000p-7, x);
        r = fma(nf, -0x1.cf79abc9e3b3ap-46, r);
        const int j = __double2loint(t) & 63;
        const int e = ((int)nf >> 6);
        const double T = lds_table[j];
        double p = 0x1.1111111111111p-7;
        p = fma(p, r, 0x1.5555555555555p-5); p = fma(p, r, 0x1.5555555555555p-3); p = fma(p, r, 0.5);
        const double q = fma(r*r, p, r);
        return __builtin_amdgcn_ldexp(fma(T, q, T), e);
    }
    // fp32: v_exp_f32 (2^p, 1 ulp) on p = x log2(e), with the rounding error of that product and the tail of log2(e) added back
    // to first order (exp2(p + d) = exp2(p)(1 + d ln 2)): 5 VALU + 1 transcendental instruction against the library's ~14
    // (its range handling: x <= 0 needs none; results below FLT_MIN flush to zero).
    __device__ __forceinline__ float exp_neg(const float x)
    {
        const float p = x * 0x1.715476p+0f;
        const float d = fmaf(x, 0x1.4ae0c0p-26f, fmaf(x, 0x1.715476p+0f, -p));
        const float y = __builtin_amdgcn_exp2f(p);
        return fmaf(y, d * 0x1.62e430p-1f, y);
    }
    __device__ __forceinline__ float exp_neg(const float x, const float*) { return exp_neg(x); }

    // sqrt(x) for normal x well inside the exponent range (here: k^2 in [1e-12, 16]): v_rsq_f64 (2^-23) + one coupled
    // Goldschmidt step + one Newton correction, i.e. the library sequence without its scaling of tiny arguments and its
    // 0 / inf / nan selects (8 instructions against about 17).
    __device__ __forceinline__ double sqrt_pos(const double x)
    {
        const double y = __builtin_amdgcn_rsq(x);
        double g = x*y, h = 0.5*y;
        const double r = fma(-h, g, 0.5);
        g = fma(g, r, g); h = fma(h, r, h);
        const double d = fma(-g, g, x);
        return fma(d, h, g);
    }
    // fp32: v_sqrt_f32 (1 ulp) without the library's scaling of denormal arguments
    __device__ __forceinline__ float sqrt_pos(const float x) { return __builtin_amdgcn_sqrtf(x); }

    // Workgroup index -> work item such that items handled by ONE XCD (its own L2) are consecutive. The dispatcher deals
    // workgroups round-robin over the eight XCDs (workgroup w runs on XCD w % 8), so neighbouring items of the natural order sit
    // behind eight different L2s; where neighbours share cache lines (a column group of 16 fp32 columns is half a 128-B line) every
    // line was then fetched twice from HBM (PMC: 2.0 x the algorithmic bytes). With this map XCD x handles items
    // [x n/8, (x+1) n/8) in dispatch order, so the two halves of a line meet in one L2 within a few workgroups of each other.
    __device__ __forceinline__ int xcd_contiguous(const int w, const int n)
    {
        constexpr int NXCD = 8;
        const int per = n / NXCD, whole = per*NXCD;
        return (w < whole) ? (w % NXCD)*per + w / NXCD : w;             // (the n % 8 last items keep their place)
    }

    inline int ceil_div(const long long a, const long long b) { return int((a + b - 1) / b); }

    // The device's default memory pool keeps what is freed into it (its release threshold is lifted once per thread and device), so
    // that the stream-ordered scratch of one solve is reused by the next without going back to the driver. Without it the pool
    // hands everything back at each synchronisation: invisible for the few KB the fused kernels ask for, 250 ms per call for the
    // per-g-point arrays of the any-nlay solver path (4 096 columns x 272 layers: 585 instead of 25 ms per step).
    // RRX_POOL_RELEASE_THRESHOLD (bytes) caps what the pool keeps when the GPU is shared with other allocators inside a host model.
    inline void keep_pool_memory()
    {
        static thread_local int configured_device = -1;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev == configured_device) return;
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool(&pool, dev) != hipSuccess) return;
        unsigned long long keep = ~0ull;
        if (const char* e = std::getenv("RRX_POOL_RELEASE_THRESHOLD")) keep = std::strtoull(e, nullptr, 10);
        if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) == hipSuccess) configured_device = dev;
    }

    // Device workspace for the large per-g-point temporaries of the any-nlay solver paths (several GB each at 4 096 columns x
    // 288 layers). Taken from the pool call by call, blocks of these sizes in changing order made hipMallocAsync go back to the
    // driver every time (rocprofv3 --hip-trace: 90 ms per call on average, 2.1 s at most; 950 ms per step where the kernels take
    // 25), so ONE grow-only block is kept per (calling thread, device, stream); work on a stream is ordered, so the block can be
    // handed out again as it is. Ownership (round 4): the block belongs to the stream. It is returned to the pool by
    // rrx_release_workspace(stream), which rrx_stream_destroy calls, and at the end of the entry point that leased it when it is
    // larger than the retention cap (RRX_WORKSPACE_KEEP bytes in the environment, default 32 GiB). An entry point leases the
    // block ONCE (WorkspaceLease) and carves everything it and the functions it calls need out of that one lease.
    // Defined in rrx_misc.hip.
    void* cached_workspace(hipStream_t st, size_t bytes);
    void release_workspace(hipStream_t st);            // the calling thread's block for this stream goes back to the pool
    void trim_workspace(hipStream_t st);               // ... only if it is larger than the retention cap
    size_t workspace_bytes(hipStream_t st);            // size of the calling thread's block for this stream (0 = none)

    class WorkspaceLease
    {
        public:
            explicit WorkspaceLease(hipStream_t st) : st_(st) {}
            WorkspaceLease(const WorkspaceLease&) = delete;
            WorkspaceLease& operator=(const WorkspaceLease&) = delete;
            // one request per lease: a second one could move the block under the first one's pointers
            template<typename F> F* get(const size_t n)
            {
                if (taken_) throw std::runtime_error("workspace leased twice in one call");
                taken_ = true;
                return static_cast<F*>(cached_workspace(st_, n*sizeof(F)));
            }
            ~WorkspaceLease() { if (taken_) trim_workspace(st_); }
        private:
            hipStream_t st_;
            bool taken_ = false;
    };

    // Stream-ordered scratch that is returned to the pool on every exit path (a throw after the first allocation must
    // not leak the earlier ones: a transient out-of-memory in a long-running host model would become permanent).
    class StreamScratch
    {
        public:
            explicit StreamScratch(hipStream_t st) : st_(st) {}
            StreamScratch(const StreamScratch&) = delete;
            StreamScratch& operator=(const StreamScratch&) = delete;
            template<typename F> F* get(const size_t n, const bool zero = false)
            {
                void* p = nullptr;
                keep_pool_memory();
                if (hipMallocAsync(&p, n*sizeof(F), st_) != hipSuccess) throw std::runtime_error("workspace allocation failed");
                ptrs_.push_back(p);
                if (zero && hipMemsetAsync(p, 0, n*sizeof(F), st_) != hipSuccess) throw std::runtime_error("workspace memset failed");
                return static_cast<F*>(p);
            }
            ~StreamScratch() { for (void* p : ptrs_) (void)hipFreeAsync(p, st_); }
        private:
            hipStream_t st_;
            std::vector<void*> ptrs_;
    };
}

namespace rrx
{
    // Number of g-point ranges the one-kernel broadband solvers split their loop into. `groups` workgroups of equal length run on
    // `slots` resident places (256 CUs x workgroups per CU), i.e. in ceil(groups / slots) rounds: few columns per GPU (BASELINE C4
    // on 8 GPUs is 2 048 each: 128 groups) leave most of the chip idle, and a column count just above a multiple of the slots
    // (16 385 columns: 1 025 groups) pays a whole extra round for its last workgroup (+20 %, round 4). Splitting the g-point loop
    // into n ranges makes n times as many workgroups of 1/n the length: the cost ceil(groups n / slots) / n is minimised over
    // n = 1, 2, 4, 8, 16 (a split must pay at least 1 % per doubling: the partial sums are written and added by a second kernel).
    // Each range sums its g-points in order into its own (nlev, ncol) partial; the partials are added in range order, so the
    // result is deterministic (it differs from the unsplit sum only in the association of the additions).
    inline int broadband_gsplit(const int groups, const int ngpt, const int slots = 512)
    {
        const Tuning& t = tuning();
        if (t.bb_gsplit >= 1) return std::min(t.bb_gsplit, ngpt);
        if (t.bb_min_groups <= 1) return 1;                               // (rrx_set_broadband_min_groups(1): never split)
        int best = 1;
        double best_cost = double(ceil_div(groups, slots));
        for (int n = 2, k = 1; n <= 16 && n*4 <= ngpt; n *= 2, ++k)
        {
            const double cost = double(ceil_div((long long)groups*n, slots)) / n * (1.0 + 0.01*k);
            if (cost < best_cost) { best_cost = cost; best = n; }
        }
        return best;
    }
}

#define RRX_TRY try {
#define RRX_CATCH(name) } catch (const std::exception& e) { rrx::set_error(std::string(name) + ": " + e.what()); return 1; } \
    return rrx::check_launch(name);

#endif
