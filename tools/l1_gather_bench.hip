// Throughput of the vector memory pipeline (TA/TCP) for L1/L2-resident gathers, per CU, as a function of the load width
// and the address pattern -- the quantity that bounds the gas-optics kernels ("gas optics roofline").
//   hipcc -O3 --offload-arch=gfx950 tools/l1_gather_bench.hip -o /tmp/l1g && /tmp/l1g
// pattern 0: all 64 lanes read the same address (a wavefront inside one LUT cell)
// pattern 1: lanes read 2-3 distinct nearby rows (neighbouring LUT cells)
// pattern 2: 64 consecutive elements (fully coalesced)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template<typename T, int PATTERN>
__global__ void __launch_bounds__(256) gather(const T* __restrict__ tab, const int nrows, const int row_elems, const int iters, T* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x*blockDim.x + threadIdx.x) >> 6;
    T acc{};
    unsigned r = wave*7u;
    for (int i=0; i<iters; i+=8)
    {
        T v[8];
        #pragma unroll
        for (int u=0; u<8; ++u)
        {
            r = r*1664525u + 1013904223u;
            const unsigned row = (r >> 8) % nrows;            // wave-uniform row: a different table row for every load
            unsigned idx;
            if (PATTERN == 0) idx = row*row_elems;
            else if (PATTERN == 1) idx = row*row_elems + (lane % 3)*14;
            else idx = row*row_elems + lane;
            v[u] = tab[idx];
        }
        #pragma unroll
        for (int u=0; u<8; ++u) acc += v[u];
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = acc;
}

struct alignas(16) D2 { double x, y; __device__ D2& operator+=(const D2& o) { x += o.x; y += o.y; return *this; } };

template<typename T, int PATTERN>
void run(const char* name, const int waves_per_cu)
{
    const int nrows = 4096, row_elems = 128;                   // 4 MB (double) / 8 MB (D2): L2 / Infinity-Cache resident
    T* tab; T* out;
    const int ncu = 256, nblk = ncu*waves_per_cu/4, iters = 4096;
    hipMalloc(&tab, sizeof(T)*nrows*row_elems); hipMemset(tab, 0, sizeof(T)*nrows*row_elems);
    hipMalloc(&out, sizeof(T)*nblk*256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    gather<T,PATTERN><<<nblk, 256>>>(tab, nrows, row_elems, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gather<T,PATTERN><<<nblk, 256>>>(tab, nrows, row_elems, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double loads_per_cu = double(iters)*waves_per_cu;
    printf("%-28s %2d waves/CU: %7.3f ms  %6.1f ns per wave-load per CU  (%5.1f clk at 2.4 GHz)  %7.1f GB/s/CU returned\n",
           name, waves_per_cu, ms, ms*1e6/loads_per_cu, ms*1e6/loads_per_cu*2.4, loads_per_cu*64*sizeof(T)/(ms*1e-3)/1e9);
    hipFree(tab); hipFree(out);
}

int main()
{
    for (int w : {8, 16})
    {
        run<double,0>("b64  same address", w);  run<D2,0>("b128 same address", w);
        run<double,1>("b64  3 nearby rows", w); run<D2,1>("b128 3 nearby rows", w);
        run<double,2>("b64  coalesced", w);     run<D2,2>("b128 coalesced", w);
    }
    return 0;
}
