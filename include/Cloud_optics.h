/* Cloud_optics_gpu -- interface of /root/reference/include/Cloud_optics.h:81-129 (LUT cloud optics, ice roughness 2) */
#ifndef CLOUD_OPTICS_H
#define CLOUD_OPTICS_H
#include "Array.h"
#include "Optical_props.h"

class Cloud_optics_gpu : public Optical_props_gpu
{
    public:
        Cloud_optics_gpu(
                const Array<Float,2>& band_lims_wvn,
                const Float radliq_lwr, const Float radliq_upr, const Float radliq_fac,
                const Float diamice_lwr, const Float diamice_upr, const Float diamice_fac,
                const Array<Float,2>& lut_extliq, const Array<Float,2>& lut_ssaliq, const Array<Float,2>& lut_asyliq,
                const Array<Float,3>& lut_extice, const Array<Float,3>& lut_ssaice, const Array<Float,3>& lut_asyice);
        void cloud_optics(
                const Array_gpu<Float,2>& clwp, const Array_gpu<Float,2>& ciwp,
                const Array_gpu<Float,2>& reliq, const Array_gpu<Float,2>& deice,
                Optical_props_1scl_gpu& optical_props);
        void cloud_optics(
                const Array_gpu<Float,2>& clwp, const Array_gpu<Float,2>& ciwp,
                const Array_gpu<Float,2>& reliq, const Array_gpu<Float,2>& deice,
                Optical_props_2str_gpu& optical_props,
                const bool delta_scale = false);      // true: optical_props.delta_scale() folded into the same pass (one kernel, same bits)
    private:
        int liq_nsteps, ice_nsteps;
        Float radliq_lwr, radliq_upr, diamice_lwr, diamice_upr;
        Array_gpu<Float,2> lut_extliq_gpu, lut_ssaliq_gpu, lut_asyliq_gpu;
        Array_gpu<Float,2> lut_extice_gpu, lut_ssaice_gpu, lut_asyice_gpu;
};
#endif
