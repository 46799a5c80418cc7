/* Gas_optics_gpu -- abstract base, interface of /root/reference/include/Gas_optics.h (GPU part) */
#ifndef GAS_OPTICS_H
#define GAS_OPTICS_H
#include <memory>
#include "Array.h"
#include "Optical_props.h"
#include "Source_functions.h"
#include "Gas_concs.h"

class Gas_optics_gpu : public Optical_props_gpu
{
    public:
        Gas_optics_gpu(const Array<Float,2>& band_lims_wvn, const Array<int,2>& band_lims_gpt) :
            Optical_props_gpu(band_lims_wvn, band_lims_gpt) {}
        virtual ~Gas_optics_gpu() {}
        virtual bool source_is_internal() const = 0;
        virtual bool source_is_external() const = 0;
        virtual Float get_press_ref_min() const = 0;
        virtual Float get_press_ref_max() const = 0;
        virtual Float get_temp_min() const = 0;
        virtual Float get_temp_max() const = 0;
        // Longwave variant.
        virtual void gas_optics(
                const Array_gpu<Float,2>& play, const Array_gpu<Float,2>& plev, const Array_gpu<Float,2>& tlay,
                const Array_gpu<Float,1>& tsfc, const Gas_concs_gpu& gas_desc,
                std::unique_ptr<Optical_props_arry_gpu>& optical_props, Source_func_lw_gpu& sources,
                const Array_gpu<Float,2>& col_dry, const Array_gpu<Float,2>& tlev) = 0;
        // Shortwave variant.
        virtual void gas_optics(
                const Array_gpu<Float,2>& play, const Array_gpu<Float,2>& plev, const Array_gpu<Float,2>& tlay,
                const Gas_concs_gpu& gas_desc,
                std::unique_ptr<Optical_props_arry_gpu>& optical_props, Array_gpu<Float,2>& toa_src,
                const Array_gpu<Float,2>& col_dry) = 0;
        virtual Float get_tsi() const = 0;
};
#endif
