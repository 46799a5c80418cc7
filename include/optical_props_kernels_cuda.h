/* Optical_props_kernels_cuda -- /root/reference/include_kernels_cuda/optical_props_kernels_cuda.h:33-56 */
#ifndef OPTICAL_PROPS_KERNELS_CUDA_H
#define OPTICAL_PROPS_KERNELS_CUDA_H
#include "rrx_forward.h"

namespace Optical_props_kernels_cuda
{
    inline void increment_1scalar_by_1scalar(int ncol, int nlay, int ngpt, Float* tau_inout, const Float* tau_in)
    { RRX_CALL(rrx_increment_1scalar_by_1scalar, ncol, nlay, ngpt, tau_inout, tau_in); }
    inline void increment_2stream_by_2stream(int ncol, int nlay, int ngpt, Float* tau_inout, Float* ssa_inout, Float* g_inout,
            const Float* tau_in, const Float* ssa_in, const Float* g_in)
    { RRX_CALL(rrx_increment_2stream_by_2stream, ncol, nlay, ngpt, tau_inout, ssa_inout, g_inout, tau_in, ssa_in, g_in); }
    inline void inc_1scalar_by_1scalar_bybnd(int ncol, int nlay, int ngpt, Float* tau_inout, const Float* tau_in, int nbnd, const int* band_lims_gpoint)
    { RRX_CALL(rrx_inc_1scalar_by_1scalar_bybnd, ncol, nlay, ngpt, tau_inout, tau_in, nbnd, band_lims_gpoint); }
    inline void inc_2stream_by_2stream_bybnd(int ncol, int nlay, int ngpt, Float* tau_inout, Float* ssa_inout, Float* g_inout,
            const Float* tau_in, const Float* ssa_in, const Float* g_in, int nbnd, const int* band_lims_gpoint)
    { RRX_CALL(rrx_inc_2stream_by_2stream_bybnd, ncol, nlay, ngpt, tau_inout, ssa_inout, g_inout, tau_in, ssa_in, g_in, nbnd, band_lims_gpoint); }
    inline void delta_scale_2str_k(int ncol, int nlay, int ngpt, Float* tau_inout, Float* ssa_inout, Float* g_inout)
    { RRX_CALL(rrx_delta_scale_2str_k, ncol, nlay, ngpt, tau_inout, ssa_inout, g_inout); }
}
#endif
