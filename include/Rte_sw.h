/* Rte_sw_gpu -- interface of /root/reference/include/Rte_sw.h:62-81 (broadband mode and a diffuse boundary condition work) */
#ifndef RTE_SW_H
#define RTE_SW_H
#include <memory>
#include "Array.h"
#include "Optical_props.h"

class Rte_sw_gpu
{
    public:
        void rte_sw(
                const std::unique_ptr<Optical_props_arry_gpu>& optical_props,
                const Bool top_at_1,
                const Array_gpu<Float,1>& mu0,
                const Array_gpu<Float,2>& inc_flux_dir,
                const Array_gpu<Float,2>& sfc_alb_dir,
                const Array_gpu<Float,2>& sfc_alb_dif,
                const Array_gpu<Float,2>& inc_flux_dif,
                Array_gpu<Float,3>& gpt_flux_up,
                Array_gpu<Float,3>& gpt_flux_dn,
                Array_gpu<Float,3>& gpt_flux_dir);
        void expand_and_transpose(
                const std::unique_ptr<Optical_props_arry_gpu>& ops,
                const Array_gpu<Float,2> arr_in,
                Array_gpu<Float,2>& arr_out);
};
#endif
