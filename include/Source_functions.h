/* Source_func_lw_gpu -- interface of /root/reference/include/Source_functions.h:66-93 */
#ifndef SOURCE_FUNCTIONS_H
#define SOURCE_FUNCTIONS_H
#include "Array.h"
#include "Optical_props.h"

class Source_func_lw_gpu : public Optical_props_gpu
{
    public:
        Source_func_lw_gpu(const int n_col, const int n_lay, const Optical_props_gpu& optical_props);
        Array_gpu<Float,2>& get_sfc_source()     { return sfc_source;     }
        Array_gpu<Float,2>& get_sfc_source_jac() { return sfc_source_jac; }
        Array_gpu<Float,3>& get_lay_source()     { return lay_source;     }
        Array_gpu<Float,3>& get_lev_source()     { return lev_source;     }
        const Array_gpu<Float,2>& get_sfc_source()     const { return sfc_source;     }
        const Array_gpu<Float,2>& get_sfc_source_jac() const { return sfc_source_jac; }
        const Array_gpu<Float,3>& get_lay_source()     const { return lay_source;     }
        const Array_gpu<Float,3>& get_lev_source()     const { return lev_source;     }
    private:
        Array_gpu<Float,2> sfc_source;
        Array_gpu<Float,2> sfc_source_jac;
        Array_gpu<Float,3> lay_source;
        Array_gpu<Float,3> lev_source;
};
#endif
