/*
 * Gas_optics_rrtmgp_gpu -- k-distribution gas optics with the constructor argument lists and methods of
 * /root/reference/include/Gas_optics_rrtmgp.h:289-408 (LW: 32 arguments, SW: 35 arguments), so that
 * load_and_init_gas_optics-style driver code (src_test/Radiation_solver.cu:70-245) compiles unchanged.
 * Construction does the one-time reductions and reorders of /root/reference/src/Gas_optics_rrtmgp.cpp:539-742
 * (drop absent gases, vmr_ref with a dry-air slot, minor-gas tables, kmajor -> (ntemp,neta,npres+1,ngpt),
 * krayl -> (ntemp,neta,ngpt,2), flavors, log-pressure grid) on the host and uploads the tables once; they stay
 * resident in HBM for the lifetime of the object.
 */
#ifndef GAS_OPTICS_RRTMGP_H
#define GAS_OPTICS_RRTMGP_H
#include <string>
#include "Gas_optics.h"

class Gas_optics_rrtmgp_gpu : public Gas_optics_gpu
{
    public:
        // Constructor for longwave variant.
        Gas_optics_rrtmgp_gpu(
                const Gas_concs_gpu& available_gases,
                const Array<std::string,1>& gas_names,
                const Array<int,3>& key_species,
                const Array<int,2>& band2gpt,
                const Array<Float,2>& band_lims_wavenum,
                const Array<Float,1>& press_ref,
                const Float press_ref_trop,
                const Array<Float,1>& temp_ref,
                const Float temp_ref_p,
                const Float temp_ref_t,
                const Array<Float,3>& vmr_ref,
                const Array<Float,4>& kmajor,
                const Array<Float,3>& kminor_lower,
                const Array<Float,3>& kminor_upper,
                const Array<std::string,1>& gas_minor,
                const Array<std::string,1>& identifier_minor,
                const Array<std::string,1>& minor_gases_lower,
                const Array<std::string,1>& minor_gases_upper,
                const Array<int,2>& minor_limits_gpt_lower,
                const Array<int,2>& minor_limits_gpt_upper,
                const Array<Bool,1>& minor_scales_with_density_lower,
                const Array<Bool,1>& minor_scales_with_density_upper,
                const Array<std::string,1>& scaling_gas_lower,
                const Array<std::string,1>& scaling_gas_upper,
                const Array<Bool,1>& scale_by_complement_lower,
                const Array<Bool,1>& scale_by_complement_upper,
                const Array<int,1>& kminor_start_lower,
                const Array<int,1>& kminor_start_upper,
                const Array<Float,2>& totplnk,
                const Array<Float,4>& planck_frac,
                const Array<Float,3>& rayl_lower,
                const Array<Float,3>& rayl_upper);

        // Constructor for shortwave variant.
        Gas_optics_rrtmgp_gpu(
                const Gas_concs_gpu& available_gases,
                const Array<std::string,1>& gas_names,
                const Array<int,3>& key_species,
                const Array<int,2>& band2gpt,
                const Array<Float,2>& band_lims_wavenum,
                const Array<Float,1>& press_ref,
                const Float press_ref_trop,
                const Array<Float,1>& temp_ref,
                const Float temp_ref_p,
                const Float temp_ref_t,
                const Array<Float,3>& vmr_ref,
                const Array<Float,4>& kmajor,
                const Array<Float,3>& kminor_lower,
                const Array<Float,3>& kminor_upper,
                const Array<std::string,1>& gas_minor,
                const Array<std::string,1>& identifier_minor,
                const Array<std::string,1>& minor_gases_lower,
                const Array<std::string,1>& minor_gases_upper,
                const Array<int,2>& minor_limits_gpt_lower,
                const Array<int,2>& minor_limits_gpt_upper,
                const Array<Bool,1>& minor_scales_with_density_lower,
                const Array<Bool,1>& minor_scales_with_density_upper,
                const Array<std::string,1>& scaling_gas_lower,
                const Array<std::string,1>& scaling_gas_upper,
                const Array<Bool,1>& scale_by_complement_lower,
                const Array<Bool,1>& scale_by_complement_upper,
                const Array<int,1>& kminor_start_lower,
                const Array<int,1>& kminor_start_upper,
                const Array<Float,1>& solar_src_quiet,
                const Array<Float,1>& solar_src_facular,
                const Array<Float,1>& solar_src_sunspot,
                const Float tsi_default,
                const Float mg_default,
                const Float sb_default,
                const Array<Float,3>& rayl_lower,
                const Array<Float,3>& rayl_upper);

        static void get_col_dry(
                Array_gpu<Float,2>& col_dry,
                const Array_gpu<Float,2>& vmr_h2o,
                const Array_gpu<Float,2>& plev);

        bool source_is_internal() const { return (totplnk.size() > 0) && (planck_frac_gpu.size() > 0); }
        bool source_is_external() const { return (solar_source.size() > 0); }
        Float get_press_ref_min() const { return press_ref_min; }
        Float get_press_ref_max() const { return press_ref_max; }
        Float get_temp_min() const { return temp_ref_min; }
        Float get_temp_max() const { return temp_ref_max; }
        int get_nflav() const { return flavor.dim(2); }
        int get_neta() const { return neta; }
        int get_npres() const { return npres; }
        int get_ntemp() const { return ntemp; }
        int get_nPlanckTemp() const { return totplnk.dim(1); }
        Float get_tsi() const;

        // Longwave variant.
        void gas_optics(
                const Array_gpu<Float,2>& play,
                const Array_gpu<Float,2>& plev,
                const Array_gpu<Float,2>& tlay,
                const Array_gpu<Float,1>& tsfc,
                const Gas_concs_gpu& gas_desc,
                std::unique_ptr<Optical_props_arry_gpu>& optical_props,
                Source_func_lw_gpu& sources,
                const Array_gpu<Float,2>& col_dry,
                const Array_gpu<Float,2>& tlev)
        { gas_optics(play, plev, tlay, tsfc, gas_desc, optical_props, sources, col_dry, tlev, nullptr); }
        // Addition: optical properties given by band (clouds) that are added to the gas optics where it is stored --
        // add_to(optical_props, *add_by_band) folded into the call, same bits, without a second pass over the g-point arrays
        void gas_optics(
                const Array_gpu<Float,2>& play, const Array_gpu<Float,2>& plev, const Array_gpu<Float,2>& tlay,
                const Array_gpu<Float,1>& tsfc, const Gas_concs_gpu& gas_desc,
                std::unique_ptr<Optical_props_arry_gpu>& optical_props, Source_func_lw_gpu& sources,
                const Array_gpu<Float,2>& col_dry, const Array_gpu<Float,2>& tlev,
                const Optical_props_1scl_gpu* add_by_band);

        // shortwave variant
        void gas_optics(
                const Array_gpu<Float,2>& play,
                const Array_gpu<Float,2>& plev,
                const Array_gpu<Float,2>& tlay,
                const Gas_concs_gpu& gas_desc,
                std::unique_ptr<Optical_props_arry_gpu>& optical_props,
                Array_gpu<Float,2>& toa_src,
                const Array_gpu<Float,2>& col_dry)
        { gas_optics(play, plev, tlay, gas_desc, optical_props, toa_src, col_dry, nullptr); }
        void gas_optics(
                const Array_gpu<Float,2>& play, const Array_gpu<Float,2>& plev, const Array_gpu<Float,2>& tlay,
                const Gas_concs_gpu& gas_desc, std::unique_ptr<Optical_props_arry_gpu>& optical_props,
                Array_gpu<Float,2>& toa_src, const Array_gpu<Float,2>& col_dry,
                const Optical_props_2str_gpu* add_by_band);

        // Extras for tests / diagnostics: the reduced gas list and host copies of the index tables.
        const Array<std::string,1>& get_gas_names() const { return gas_names; }
        const Array<int,2>& get_flavor() const { return flavor; }
        const Array<int,2>& get_gpoint_flavor() const { return gpoint_flavor; }

    private:
        int ntemp = 0, neta = 0, npres = 0;
        Float totplnk_delta = 0, temp_ref_min = 0, temp_ref_max = 0, press_ref_min = 0, press_ref_max = 0;
        Float press_ref_trop_log = 0, press_ref_log_delta = 0, temp_ref_delta = 0;
        int idx_h2o = -1;
        bool has_rayleigh = false;

        Array<std::string,1> gas_names;
        Array<int,2> flavor, gpoint_flavor;
        Array<Float,2> totplnk;
        Array<Float,1> solar_source_quiet, solar_source_facular, solar_source_sunspot, solar_source;
        int nminorlower = 0, nminorklower = 0, nminorupper = 0, nminorkupper = 0;

        Array_gpu<Float,1> press_ref_log_gpu, temp_ref_gpu, solar_source_gpu;
        Array_gpu<Float,3> vmr_ref_gpu;
        Array_gpu<int,2> flavor_gpu, gpoint_flavor_gpu;
        Array_gpu<Float,4> kmajor_gpu, planck_frac_gpu, krayl_gpu;
        Array_gpu<Float,2> totplnk_gpu;
        Array_gpu<Float,3> kminor_lower_gpu, kminor_upper_gpu;
        Array_gpu<int,2> minor_limits_gpt_lower_gpu, minor_limits_gpt_upper_gpu;
        Array_gpu<Bool,1> minor_scales_with_density_lower_gpu, minor_scales_with_density_upper_gpu;
        Array_gpu<Bool,1> scale_by_complement_lower_gpu, scale_by_complement_upper_gpu;
        Array_gpu<int,1> kminor_start_lower_gpu, kminor_start_upper_gpu;
        Array_gpu<int,1> idx_minor_lower_gpu, idx_minor_upper_gpu, idx_minor_scaling_lower_gpu, idx_minor_scaling_upper_gpu;

        void init_abs_coeffs(
                const Gas_concs_gpu& available_gases,
                const Array<std::string,1>& gas_names,
                const Array<int,3>& key_species,
                const Array<Float,1>& press_ref,
                const Array<Float,1>& temp_ref,
                const Float press_ref_trop,
                const Array<Float,3>& vmr_ref,
                const Array<Float,4>& kmajor,
                const Array<Float,3>& kminor_lower,
                const Array<Float,3>& kminor_upper,
                const Array<std::string,1>& gas_minor,
                const Array<std::string,1>& identifier_minor,
                const Array<std::string,1>& minor_gases_lower,
                const Array<std::string,1>& minor_gases_upper,
                const Array<int,2>& minor_limits_gpt_lower,
                const Array<int,2>& minor_limits_gpt_upper,
                const Array<Bool,1>& minor_scales_with_density_lower,
                const Array<Bool,1>& minor_scales_with_density_upper,
                const Array<std::string,1>& scaling_gas_lower,
                const Array<std::string,1>& scaling_gas_upper,
                const Array<Bool,1>& scale_by_complement_lower,
                const Array<Bool,1>& scale_by_complement_upper,
                const Array<int,1>& kminor_start_lower,
                const Array<int,1>& kminor_start_upper,
                const Array<Float,3>& rayl_lower,
                const Array<Float,3>& rayl_upper);

        void set_solar_variability(const Float mg_index, const Float sb_index);

        void fill_col_gas(const int ncol, const int nlay, const Gas_concs_gpu& gas_desc, const Array_gpu<Float,2>& col_dry,
                          Array_gpu<Float,3>& col_gas);
        int vertical_ordering = -1;
    public:
        // Host-model coupling: -1 (default) = detect the vertical ordering from play with the reference's synchronous
        // one-element read-back (src_cuda/Gas_optics_rrtmgp.cu:1190); 0 = surface first, 1 = top of atmosphere first:
        // no read-back, gas_optics() is then fully asynchronous on the calling thread's stream (rrx_host::set_stream).
        void set_vertical_ordering(const int top_at_1) { vertical_ordering = top_at_1; }
};
#endif
