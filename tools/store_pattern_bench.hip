// Store-pattern probe for the windowed gas optics (round 4): how fast does MI355X take the kernel's WRITE pattern by itself?
// A workgroup of 256 threads owns CPT*256 columns of one layer and walks over all g-points, storing NARR arrays of the
// (col, lay, gpt) layout -- exactly gas_window_kernel's stores (geometry 256 x 1, non-temporal) without its loads and arithmetic.
// Variants: cells per thread (8 / 16 / 32 B per lane and store), plain vs non-temporal stores, g-points in chunks of 16 with a
// pause between chunks (the kernel's bursts). Build: hipcc --offload-arch=gfx950 -O3. Output: TB/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template<typename T, bool NT>
__device__ __forceinline__ void st(T* p, const T v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template<typename E, int CPT, int NARR, bool NT, int PAUSE>
__global__ void __launch_bounds__(256, 3) writer(const int ncol, const int nlay, const int ngpt, E* __restrict__ out, const E seed)
{
    typedef E V __attribute__((ext_vector_type(CPT)));
    const int icol = (blockIdx.x*256 + threadIdx.x)*CPT;
    const int ilay = blockIdx.y;
    if (icol >= ncol) return;
    const size_t ncl = size_t(ncol)*nlay;
    const size_t idx = icol + size_t(ilay)*ncol;
    E v = seed + icol;
    for (int c0=0; c0<ngpt; c0+=16)
    {
        if (PAUSE > 0) { for (int k=0; k<PAUSE; ++k) v = v * E(1.0000001) + E(1e-9); __builtin_amdgcn_s_sleep(PAUSE > 64 ? 64 : PAUSE); }
        #pragma unroll 4
        for (int ig=c0; ig<min(c0+16, ngpt); ++ig)
        {
            v = v * E(1.0000001) + E(1e-9);
            #pragma unroll
            for (int a=0; a<NARR; ++a)
            {
                V x;
                #pragma unroll
                for (int k=0; k<CPT; ++k) x[k] = v + a + k;
                if constexpr (CPT == 1) st<E,NT>(out + size_t(a)*ncl*ngpt + size_t(ig)*ncl + idx, x[0]);
                else st<V,NT>(reinterpret_cast<V*>(out + size_t(a)*ncl*ngpt + size_t(ig)*ncl + idx), x);
            }
        }
    }
}

template<typename E, int CPT, int NARR, bool NT, int PAUSE>
void run(const char* name, int ncol, int nlay, int ngpt, void* out_)
{
    E* out = static_cast<E*>(out_);
    const dim3 grid((ncol/CPT + 255)/256, nlay);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    writer<E,CPT,NARR,NT,PAUSE><<<grid, 256>>>(ncol, nlay, ngpt, out, E(1.5));
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r=0; r<5; ++r) writer<E,CPT,NARR,NT,PAUSE><<<grid, 256>>>(ncol, nlay, ngpt, out, E(1.5 + r));
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double gb = double(ncol)*nlay*ngpt*NARR*sizeof(E)/1e9;
    std::printf("%-44s %6.2f GB  %7.3f ms  %5.2f TB/s\n", name, gb, ms, gb/ms);
}

int main()
{
    const int ncol = 16384, nlay = 140, ngpt = 256;
    double* out; hipMalloc(&out, size_t(ncol)*nlay*ngpt*2*8);
    run<double,1,2,true,0>("1 cell/thread, 2 arrays, nt", ncol, nlay, ngpt, out);
    run<double,1,2,false,0>("1 cell/thread, 2 arrays, plain", ncol, nlay, ngpt, out);
    run<double,2,2,true,0>("2 cells/thread (16 B), 2 arrays, nt", ncol, nlay, ngpt, out);
    run<double,4,2,true,0>("4 cells/thread (32 B), 2 arrays, nt", ncol, nlay, ngpt, out);
    run<double,1,1,true,0>("1 cell/thread, 1 array, nt", ncol, nlay, ngpt, out);
    run<double,1,2,true,200>("1 cell/thread, 2 arrays, nt, pause 200", ncol, nlay, ngpt, out);
    run<double,1,2,true,1000>("1 cell/thread, 2 arrays, nt, pause 1000", ncol, nlay, ngpt, out);
    run<double,2,2,true,1000>("2 cells/thread, 2 arrays, nt, pause 1000", ncol, nlay, ngpt, out);
    // fp32 (round 4, late): the same pattern with 4-B elements -- one cell per lane is a 256-B store per wavefront
    run<float,1,2,true,0>("fp32 1 cell/thread (4 B), 2 arrays, nt", ncol, nlay, ngpt, out);
    run<float,2,2,true,0>("fp32 2 cells/thread (8 B), 2 arrays, nt", ncol, nlay, ngpt, out);
    run<float,4,2,true,0>("fp32 4 cells/thread (16 B), 2 arrays, nt", ncol, nlay, ngpt, out);
    run<float,1,3,true,0>("fp32 1 cell/thread, 3 arrays, nt", ncol, nlay, ngpt, out);
    run<float,2,3,true,0>("fp32 2 cells/thread, 3 arrays, nt", ncol, nlay, ngpt, out);
    run<float,1,2,true,1000>("fp32 1 cell/thread, 2 arrays, nt, pause 1000", ncol, nlay, ngpt, out);
    run<float,2,2,true,1000>("fp32 2 cells/thread, 2 arrays, nt, pause 1000", ncol, nlay, ngpt, out);
    return 0;
}
