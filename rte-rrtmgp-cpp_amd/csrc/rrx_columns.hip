// Column ordering for the product chain (round 4; no counterpart in the reference library): columns are independent, so a solve
// may process them in any order. Two uses, one mechanism -- a gather index `perm` of length ncol + npad:
//   * order: ascending surface pressure. The windowed gas optics stages one box of LUT nodes per 256 neighbouring cells; columns
//     that differ by more than about one cell of the LUT's pressure grid do not fit a box and are handed back to the gather kernels
//     (+60 % per step at +-35 % pressure spread). Sorted, neighbours are alike again.
//   * padding: perm[ncol .. ncol+npad) repeats the last column, so that the solve runs on a multiple of 16 columns (rows of the
//     (col, lay, gpt) arrays then start on 128-B lines: 16 385 columns cost 20 % more than 16 384 unpadded).
// Inputs are gathered through perm, outputs scattered back through its first ncol entries. Pure data movement + one radix sort.
#include "rrx_common.h"
#include "rrx_hip.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace
{
using namespace rrx;

__global__ void iota_pad_kernel(const int ncol, const int npad, int* __restrict__ perm)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i < ncol + npad) perm[i] = min(i, ncol-1);
}

__global__ void pad_perm_kernel(const int ncol, const int npad, int* __restrict__ perm)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i < npad) perm[ncol + i] = perm[ncol-1];
}

// flag = 1 where some run of `block` consecutive columns spans more than `threshold` of its mean (one workgroup per run)
template<typename F>
__global__ void __launch_bounds__(256) column_spread_kernel(const int ncol, const F* __restrict__ key, const int block, const F threshold, int* __restrict__ flag)
{
    __shared__ F s_min[256], s_max[256], s_sum[256];
    const int c0 = blockIdx.x*block;
    F lo = (sizeof(F) == 8) ? F(1e300) : F(3e38), hi = -lo, sum = F(0.);
    for (int i = c0 + threadIdx.x; i < min(c0 + block, ncol); i += 256) { const F v = key[i]; lo = min(lo, v); hi = max(hi, v); sum += v; }
    s_min[threadIdx.x] = lo; s_max[threadIdx.x] = hi; s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int s=128; s>0; s>>=1)
    {
        if (int(threadIdx.x) < s)
        {
            s_min[threadIdx.x] = min(s_min[threadIdx.x], s_min[threadIdx.x+s]); s_max[threadIdx.x] = max(s_max[threadIdx.x], s_max[threadIdx.x+s]);
            s_sum[threadIdx.x] += s_sum[threadIdx.x+s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        const int n = min(c0 + block, ncol) - c0;
        if (n == block && (s_max[0] - s_min[0]) > threshold * (s_sum[0] / F(n))) atomicExch(flag, 1);
    }
}

// out(i, r) = in(perm[i], r): arrays whose FIRST (fastest) dimension is the column
template<typename T>
__global__ void gather_cols_kernel(const int nout, const size_t nrest, const int* __restrict__ perm, const int ncol_in, const T* __restrict__ in, T* __restrict__ out)
{
    const size_t r = blockIdx.y;
    for (int i = blockIdx.x*blockDim.x + threadIdx.x; i < nout; i += gridDim.x*blockDim.x)
        for (size_t rr = r; rr < nrest; rr += gridDim.y) out[i + rr*nout] = in[perm[i] + rr*ncol_in];
}

// out(perm[i], r) = in(i, r), i < n: the inverse, into an array of ncol_dst columns from one of ncol_src
template<typename T>
__global__ void scatter_cols_kernel(const int n, const size_t nrest, const int* __restrict__ perm, const int ncol_src, const T* __restrict__ in, const int ncol_dst, T* __restrict__ out)
{
    const size_t r = blockIdx.y;
    for (int i = blockIdx.x*blockDim.x + threadIdx.x; i < n; i += gridDim.x*blockDim.x)
        for (size_t rr = r; rr < nrest; rr += gridDim.y) out[perm[i] + rr*ncol_dst] = in[i + rr*ncol_src];
}

// out(b, i) = in(b, perm[i]): arrays whose LAST dimension is the column, e.g. emis_sfc(nbnd, ncol)
template<typename T>
__global__ void gather_lastdim_kernel(const int n1, const int nout, const int* __restrict__ perm, const T* __restrict__ in, T* __restrict__ out)
{
    const size_t n = size_t(n1)*nout;
    for (size_t k = size_t(blockIdx.x)*blockDim.x + threadIdx.x; k < n; k += size_t(gridDim.x)*blockDim.x)
    {
        const int b = int(k % n1), i = int(k / n1);
        out[k] = in[b + size_t(perm[i])*n1];
    }
}

inline dim3 grid2(const int n, const size_t nrest) { return dim3(std::min(ceil_div(n, 256), 256), unsigned(std::min<size_t>(nrest, 4096))); }

template<typename F>
int sort_columns_impl(const int ncol, const F* key, const int npad, int* perm, void* stream)
{
    RRX_TRY
    if (ncol <= 0 || npad < 0) throw std::runtime_error("empty problem");
    hipStream_t st = static_cast<hipStream_t>(stream);
    StreamScratch scratch(st);
    int* iota = scratch.get<int>(size_t(ncol));
    F* keys_out = scratch.get<F>(size_t(ncol));
    iota_pad_kernel<<<ceil_div(ncol, 256), 256, 0, st>>>(ncol, 0, iota);
    size_t temp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, temp_bytes, key, keys_out, iota, perm, size_t(ncol), 0, 8*sizeof(F), st) != hipSuccess)
        throw std::runtime_error("radix sort set-up failed");
    void* temp = scratch.get<char>(std::max<size_t>(temp_bytes, 16));
    if (rocprim::radix_sort_pairs(temp, temp_bytes, key, keys_out, iota, perm, size_t(ncol), 0, 8*sizeof(F), st) != hipSuccess)
        throw std::runtime_error("radix sort failed");
    if (npad > 0) pad_perm_kernel<<<ceil_div(npad, 256), 256, 0, st>>>(ncol, npad, perm);
    RRX_CATCH("rrx_sort_columns")
}
}  // namespace


extern "C"
{
int rrx_identity_columns(int ncol, int npad, int* perm, void* stream)
{
    RRX_TRY
    if (ncol <= 0 || npad < 0) throw std::runtime_error("empty problem");
    iota_pad_kernel<<<rrx::ceil_div(ncol + npad, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(ncol, npad, perm);
    RRX_CATCH("rrx_identity_columns")
}

#define RRX_DEFINE_COLUMNS(F, SFX) \
int rrx_sort_columns##SFX(int ncol, const F* key, int npad, int* perm, void* stream) { return sort_columns_impl<F>(ncol, key, npad, perm, stream); } \
int rrx_column_spread##SFX(int ncol, const F* key, int block, F threshold, int* flag, void* stream) \
{ RRX_TRY if (ncol <= 0 || block <= 0) throw std::runtime_error("empty problem"); \
  hipStream_t st = static_cast<hipStream_t>(stream); \
  if (hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) throw std::runtime_error("memset failed"); \
  column_spread_kernel<F><<<rrx::ceil_div(ncol, block), 256, 0, st>>>(ncol, key, block, threshold, flag); RRX_CATCH("rrx_column_spread") } \
int rrx_gather_cols##SFX(int nout, unsigned long long nrest, const int* perm, int ncol_in, const F* in, F* out, void* stream) \
{ RRX_TRY if (nout <= 0 || nrest == 0) return 0; \
  gather_cols_kernel<F><<<grid2(nout, nrest), 256, 0, static_cast<hipStream_t>(stream)>>>(nout, size_t(nrest), perm, ncol_in, in, out); RRX_CATCH("rrx_gather_cols") } \
int rrx_scatter_cols##SFX(int n, unsigned long long nrest, const int* perm, int ncol_src, const F* in, int ncol_dst, F* out, void* stream) \
{ RRX_TRY if (n <= 0 || nrest == 0) return 0; \
  scatter_cols_kernel<F><<<grid2(n, nrest), 256, 0, static_cast<hipStream_t>(stream)>>>(n, size_t(nrest), perm, ncol_src, in, ncol_dst, out); RRX_CATCH("rrx_scatter_cols") } \
int rrx_gather_lastdim##SFX(int n1, int nout, const int* perm, const F* in, F* out, void* stream) \
{ RRX_TRY if (n1 <= 0 || nout <= 0) return 0; \
  gather_lastdim_kernel<F><<<rrx::ceil_div(size_t(n1)*nout, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(n1, nout, perm, in, out); RRX_CATCH("rrx_gather_lastdim") }

RRX_DEFINE_COLUMNS(double, _f64)
RRX_DEFINE_COLUMNS(float, _f32)
}
