/*
 * TEST INFRASTRUCTURE ONLY (oracle/): never linked into, imported by or called from the product path.
 *
 * Execution-model shim that lets the reference's CUDA kernel *text*
 * (/root/reference/src_kernels_cuda/{rte_solver,gas_optics_rrtmgp,optical_props,fluxes}_kernels.cu)
 * be compiled by g++ and executed sequentially on the host, read in place from /root/reference
 * (nothing from the reference is copied into this repository).
 *
 * What is provided here is ONLY the CUDA execution model (qualifiers, dim3, the built-in index
 * variables and a sequential "launch"); no reference header, library, tool or generated code is
 * replaced: all arithmetic comes from the reference kernel text itself.
 *
 * SURVEY.md section 8(c) describes the approach and its one caveat: gas_optical_depths_minor_kernel uses a
 * __shared__ broadcast between threadIdx.x lanes; it is instantiated as <1,1,1> so that threadIdx.x == 0
 * for every emulated thread and the sequential emulation is exact.
 */
#ifndef RRX_ORACLE_CUDA_ON_HOST_H
#define RRX_ORACLE_CUDA_ON_HOST_H

#include <cmath>
#include <cfloat>
#include <algorithm>
#include <limits>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __shared__ static thread_local

struct dim3
{
    unsigned int x, y, z;
    dim3(unsigned int x_=1, unsigned int y_=1, unsigned int z_=1) : x(x_), y(y_), z(z_) {}
};

static thread_local dim3 blockIdx, threadIdx, blockDim(1, 1, 1), gridDim(1, 1, 1);

inline void __syncthreads() {}

using std::min;
using std::max;
using std::abs;
using std::exp;
using std::log;
using std::sqrt;
using std::fmod;
using std::acos;

// Sequential launch: one emulated thread per grid cell, blockDim = (1,1,1).
template<class Kernel, class... Args>
inline void host_launch(const dim3 grid, Kernel kernel, Args... args)
{
    blockDim = dim3(1, 1, 1);
    gridDim = grid;
    threadIdx = dim3(0, 0, 0);
    for (unsigned int z=0; z<grid.z; ++z)
        for (unsigned int y=0; y<grid.y; ++y)
            for (unsigned int x=0; x<grid.x; ++x)
            {
                blockIdx = dim3(x, y, z);
                kernel(args...);
            }
}

#endif
