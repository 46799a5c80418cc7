/* Aerosol_optics_gpu -- interface of /root/reference/include/Aerosol_optics.h:19-20,51-86 (CAMS aerosol optics: 11 species,
 * hydrophilic ones by humidity class). Aerosol concentrations travel in a Gas_concs_gpu under the names aermr01..aermr11, each a
 * (ncol, nlay) field or a (1, nlay) profile. */
#ifndef AEROSOL_OPTICS_H
#define AEROSOL_OPTICS_H
#include "Array.h"
#include "Optical_props.h"
#include "Gas_concs.h"

using Aerosol_concs = Gas_concs;
using Aerosol_concs_gpu = Gas_concs_gpu;

class Aerosol_optics_gpu : public Optical_props_gpu
{
    public:
        // tables as the reference's loader hands them over: hydrophobic (n_band, n_phobic), hydrophilic (n_band, n_hum, n_philic)
        Aerosol_optics_gpu(
                const Array<Float,2>& band_lims_wvn, const Array<Float,1>& rh_upper,
                const Array<Float,2>& mext_phobic, const Array<Float,2>& ssa_phobic, const Array<Float,2>& g_phobic,
                const Array<Float,3>& mext_philic, const Array<Float,3>& ssa_philic, const Array<Float,3>& g_philic);

        // Profiles in aerosol_concs are read in place by the kernel (the reference first broadcasts them to (ncol, nlay) and
        // stores them back, which is why its argument is not const; kept for signature compatibility)
        void aerosol_optics(
                Aerosol_concs_gpu& aerosol_concs,
                const Array_gpu<Float,2>& rh, const Array_gpu<Float,2>& plev,
                Optical_props_2str_gpu& optical_props);

    private:
        int n_hum, n_phobic, n_philic;
        Array_gpu<Float,1> rh_upper_gpu;
        Array_gpu<Float,2> mext_phobic_gpu, ssa_phobic_gpu, g_phobic_gpu;
        Array_gpu<Float,3> mext_philic_gpu, ssa_philic_gpu, g_philic_gpu;
};
#endif
