// rrx_rccl.hip -- all-gather of column-sharded flux arrays over RCCL; see include/rrx_rccl.h.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <unistd.h>
#include "rrx_rccl.h"

namespace
{
    thread_local std::string last_error;

    struct Comm { ncclComm_t nccl; int world, rank; };

    void check(ncclResult_t r, const char* what)
    {
        if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
    }
    void check(hipError_t r, const char* what)
    {
        if (r != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(r));
    }

    // local (nrows, nloc) -> padded (nrows, nmax); columns beyond nloc are never read back
    template<typename F>
    __global__ void pad_columns_kernel(const size_t n, const int nloc, const int nmax, const F* __restrict__ in, F* __restrict__ out)
    {
        for (size_t i = blockIdx.x*size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x)*blockDim.x)
        {
            const size_t row = i / nloc; const int c = int(i - row*nloc);
            out[row*nmax + c] = in[i];
        }
    }

    // recv (world, nrows, nmax) -> gathered (nrows, ntot): rank r's columns go to [start_r, start_r + n_r)
    template<typename F>
    __global__ void place_columns_kernel(const size_t n, const int ntot, const int nmax, const int world, const int nrows,
                                         const F* __restrict__ recv, F* __restrict__ out)
    {
        const int base = ntot / world, extra = ntot % world;
        for (size_t i = blockIdx.x*size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x)*blockDim.x)
        {
            const size_t row = i / ntot; const int c = int(i - row*ntot);
            // owner of global column c: the first `extra` ranks hold base+1 columns
            const int split = extra*(base + 1);
            const int r = (c < split) ? c / (base + 1) : extra + (c - split) / max(base, 1);
            const int start = r*base + min(r, extra);
            out[i] = recv[(size_t(r)*nrows + row)*nmax + (c - start)];
        }
    }

    template<typename F>
    int allgather(void* comm_, const int nrows, const int ntot, const F* local, F* gathered, F* scratch, void* stream_, ncclDataType_t dt)
    {
        try
        {
            Comm* c = static_cast<Comm*>(comm_);
            if (!c) throw std::runtime_error("no communicator");
            if (nrows < 1 || ntot < c->world) throw std::runtime_error("rrx_allgather_fluxes: fewer columns than ranks");
            hipStream_t st = static_cast<hipStream_t>(stream_);
            int s, e; rrx_column_range(c->rank, c->world, ntot, &s, &e);
            const int nloc = e - s, nmax = (ntot + c->world - 1) / c->world;
            const size_t blk = size_t(nrows)*nmax;
            const F* send = local;
            if (nloc != nmax)
            {
                const size_t n = size_t(nrows)*nloc;
                pad_columns_kernel<F><<<int(std::min<size_t>((n + 255)/256, 4096)), 256, 0, st>>>(n, nloc, nmax, local, scratch);
                send = scratch;
            }
            F* recv = scratch + blk;
            check(ncclAllGather(send, recv, blk, dt, c->nccl, st), "ncclAllGather");
            const size_t n = size_t(nrows)*ntot;
            place_columns_kernel<F><<<int(std::min<size_t>((n + 255)/256, 8192)), 256, 0, st>>>(n, ntot, nmax, c->world, nrows, recv, gathered);
            check(hipGetLastError(), "place_columns_kernel");
            return 0;
        }
        catch (const std::exception& ex) { last_error = ex.what(); return 1; }
    }
}

extern "C" {

const char* rrx_rccl_last_error(void) { return last_error.c_str(); }

void rrx_column_range(int rank, int world, int ncol_total, int* col_s, int* col_e)
{
    const int base = ncol_total / world, extra = ncol_total % world;
    *col_s = rank*base + (rank < extra ? rank : extra);
    *col_e = *col_s + base + (rank < extra ? 1 : 0);
}

int rrx_comm_get_unique_id(char id[RRX_COMM_ID_BYTES])
{
    try
    {
        static_assert(sizeof(ncclUniqueId) <= RRX_COMM_ID_BYTES, "ncclUniqueId larger than RRX_COMM_ID_BYTES");
        ncclUniqueId u; check(ncclGetUniqueId(&u), "ncclGetUniqueId");
        std::memset(id, 0, RRX_COMM_ID_BYTES); std::memcpy(id, &u, sizeof(u));
        return 0;
    }
    catch (const std::exception& ex) { last_error = ex.what(); return 1; }
}

int rrx_comm_id_to_file(const char* path, const char id[RRX_COMM_ID_BYTES])
{
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(id, 1, RRX_COMM_ID_BYTES, f) != RRX_COMM_ID_BYTES) { last_error = "cannot write " + tmp; if (f) std::fclose(f); return 1; }
    std::fclose(f);
    if (std::rename(tmp.c_str(), path) != 0) { last_error = std::string("cannot publish ") + path; return 1; }
    return 0;
}

int rrx_comm_id_from_file(const char* path, char id[RRX_COMM_ID_BYTES], int timeout_s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;)
    {
        FILE* f = std::fopen(path, "rb");
        if (f)
        {
            const size_t n = std::fread(id, 1, RRX_COMM_ID_BYTES, f);
            std::fclose(f);
            if (n == RRX_COMM_ID_BYTES) return 0;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s))
        { last_error = std::string("timed out waiting for ") + path; return 1; }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
}

int rrx_comm_create(int world, int rank, const char id[RRX_COMM_ID_BYTES], void** comm)
{
    try
    {
        if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("rrx_comm_create: bad rank / world size");
        ncclUniqueId u; std::memcpy(&u, id, sizeof(u));
        Comm* c = new Comm{nullptr, world, rank};
        const ncclResult_t r = ncclCommInitRank(&c->nccl, world, u, rank);
        if (r != ncclSuccess) { delete c; check(r, "ncclCommInitRank"); }
        *comm = c;
        return 0;
    }
    catch (const std::exception& ex) { last_error = ex.what(); return 1; }
}

int rrx_comm_destroy(void* comm)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return 0;
    const ncclResult_t r = ncclCommDestroy(c->nccl);
    delete c;
    if (r != ncclSuccess) { last_error = ncclGetErrorString(r); return 1; }
    return 0;
}

// Layout check without a second GPU: shards a known (nrows, ncol_total) array as `world` ranks would hold it, pads each shard,
// lays the blocks out as ncclAllGather delivers them and runs the placing kernel. Returns 0 when the result is the original.
int rrx_rccl_selftest_layout(int world, int nrows, int ncol_total)
{
    try
    {
        const int nmax = (ncol_total + world - 1) / world;
        const size_t blk = size_t(nrows)*nmax, n = size_t(nrows)*ncol_total;
        std::string host(n*sizeof(double), 0);
        double* h = reinterpret_cast<double*>(&host[0]);
        for (size_t i=0; i<n; ++i) h[i] = 1.0 + 0.25*double(i);
        double *full, *recv, *out, *loc;
        check(hipMalloc(&full, n*8), "hipMalloc"); check(hipMalloc(&recv, world*blk*8), "hipMalloc");
        check(hipMalloc(&out, n*8), "hipMalloc"); check(hipMalloc(&loc, blk*8), "hipMalloc");
        check(hipMemset(recv, 0xff, world*blk*8), "hipMemset");
        for (int r=0; r<world; ++r)
        {
            int s, e; rrx_column_range(r, world, ncol_total, &s, &e);
            const int nloc = e - s;
            for (int row=0; row<nrows; ++row)          // the rank's local array (nrows, nloc)
                check(hipMemcpy(loc + size_t(row)*nloc, h + size_t(row)*ncol_total + s, nloc*8, hipMemcpyHostToDevice), "hipMemcpy");
            const size_t nl = size_t(nrows)*nloc;
            if (nloc != nmax) pad_columns_kernel<double><<<64, 256>>>(nl, nloc, nmax, loc, recv + r*blk);
            else check(hipMemcpy(recv + r*blk, loc, blk*8, hipMemcpyDeviceToDevice), "hipMemcpy");
        }
        place_columns_kernel<double><<<64, 256>>>(n, ncol_total, nmax, world, nrows, recv, out);
        std::string back(n*sizeof(double), 0);
        check(hipMemcpy(&back[0], out, n*8, hipMemcpyDeviceToHost), "hipMemcpy");
        (void)hipFree(full); (void)hipFree(recv); (void)hipFree(out); (void)hipFree(loc);
        return std::memcmp(back.data(), host.data(), n*8) == 0 ? 0 : 2;
    }
    catch (const std::exception& ex) { last_error = ex.what(); return 1; }
}

int rrx_allgather_fluxes_f64(void* comm, int nrows, int ncol_total, const double* local, double* gathered, double* scratch, void* stream)
{ return allgather<double>(comm, nrows, ncol_total, local, gathered, scratch, stream, ncclDouble); }
int rrx_allgather_fluxes_f32(void* comm, int nrows, int ncol_total, const float* local, float* gathered, float* scratch, void* stream)
{ return allgather<float>(comm, nrows, ncol_total, local, gathered, scratch, stream, ncclFloat); }

}
