/* Rte_lw_gpu -- interface of /root/reference/include/Rte_lw.h:62-80. Unlike the reference GPU class, broadband mode
 * (gpt_flux arrays with third dimension 1, as the CPU path uses: src/Rte_lw.cpp:176) and n_gauss_angles 1..4 work. */
#ifndef RTE_LW_H
#define RTE_LW_H
#include <memory>
#include "Array.h"
#include "Optical_props.h"
#include "Source_functions.h"

class Rte_lw_gpu
{
    public:
        void rte_lw(
                const std::unique_ptr<Optical_props_arry_gpu>& optical_props,
                const Bool top_at_1,
                const Source_func_lw_gpu& sources,
                const Array_gpu<Float,2>& sfc_emis,
                const Array_gpu<Float,2>& inc_flux,
                Array_gpu<Float,3>& gpt_flux_up,
                Array_gpu<Float,3>& gpt_flux_dn,
                const int n_gauss_angles);
        void expand_and_transpose(
                const std::unique_ptr<Optical_props_arry_gpu>& ops,
                const Array_gpu<Float,2> arr_in,
                Array_gpu<Float,2>& arr_out);
    private:
        // Gauss-Jacobi secants and weights on the device, uploaded once per object and angle count (an upload per call is a host
        // copy the stream is synchronised for: the solver launch then waits for the gas optics to finish before it is even enqueued)
        Array_gpu<Float,2> gauss_Ds_gpu, gauss_wts_gpu;
        int gauss_angles_cached = 0;
};
#endif
