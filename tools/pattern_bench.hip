// Access-pattern probe for the solver tiling: how fast can MI355X stream (col,lay,gpt) arrays when a wave touches
// CL columns x (64/CL) level rows per instruction (segments of CL*8 bytes, row stride ncol*8 bytes)?
// Pure data movement, 3 reads + 2 writes per cell like lw_solver_noscat. Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template<int CL, int UNROLL>
__global__ void __launch_bounds__(256) tile_copy(const int ncol, const int nlay, const int ngpt,
        const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c,
        double* __restrict__ o1, double* __restrict__ o2)
{
    constexpr int LL = 64 / CL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane % CL, ll = lane / CL;
    const int icol = (blockIdx.x*4 + wave)*CL + cl;
    const int igpt = blockIdx.y;
    if (icol >= ncol) return;
    const int K = (nlay + LL - 1) / LL;
    const size_t base = size_t(igpt)*ncol*nlay + icol;
    for (int j0=0; j0<K; j0+=UNROLL)
    {
        double va[UNROLL], vb[UNROLL], vc[UNROLL];
        #pragma unroll
        for (int u=0; u<UNROLL; ++u)
        {
            const int t = min(ll*K + j0 + u, nlay-1);
            va[u] = a[base + size_t(t)*ncol]; vb[u] = b[base + size_t(t)*ncol]; vc[u] = c[base + size_t(t)*ncol];
        }
        #pragma unroll
        for (int u=0; u<UNROLL; ++u)
        {
            const int t = ll*K + j0 + u;
            if (j0 + u < K && t < nlay) { o1[base + size_t(t)*ncol] = va[u] + vb[u]; o2[base + size_t(t)*ncol] = vc[u] * 2.0; }
        }
    }
}

template<int CL, int UNROLL>
float run(int ncol, int nlay, int ngpt, double* a, double* b, double* c, double* o1, double* o2)
{
    dim3 grid((ncol + 4*CL - 1)/(4*CL), ngpt);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    tile_copy<CL,UNROLL><<<grid, 256>>>(ncol, nlay, ngpt, a, b, c, o1, o2);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r=0; r<3; ++r)
    {
        hipEventRecord(e0);
        tile_copy<CL,UNROLL><<<grid, 256>>>(ncol, nlay, ngpt, a, b, c, o1, o2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int ncol = 16384, nlay = 140, ngpt = 256;
    const size_t n = size_t(ncol)*nlay*ngpt;
    double *a, *b, *c, *o1, *o2;
    hipMalloc(&a, n*8); hipMalloc(&b, n*8); hipMalloc(&c, n*8); hipMalloc(&o1, n*8); hipMalloc(&o2, n*8);
    hipMemset(a, 0, n*8); hipMemset(b, 0, n*8); hipMemset(c, 0, n*8);
    const double gb = 5.0*n*8/1e9;
    #define RUN(CL, U) { float ms = run<CL,U>(ncol, nlay, ngpt, a, b, c, o1, o2); \
        printf("segment %4d B (CL=%2d) unroll %2d : %7.3f ms  %7.1f GB/s\n", CL*8, CL, U, ms, gb/ms*1e3); }
    RUN(8, 6) RUN(8, 18) RUN(16, 6) RUN(16, 12) RUN(32, 6) RUN(32, 12) RUN(64, 4) RUN(64, 12) RUN(4, 9)
    return 0;
}
