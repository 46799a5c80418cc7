// Radiation_solver_longwave / _shortwave (GPU path): load the k-distribution and cloud LUT, then run
// gas optics -> [clouds] -> solver -> flux reduction per column block, scattering block results into the full arrays.
// Flow of /root/reference/src_test/Radiation_solver.cu:405-950; coefficient loading as :70-357 of that file.
#include <algorithm>
#include <numeric>
#include "Radiation_solver.h"
#include "Netcdf_interface.h"
#include "subset_kernels_cuda.h"
#include "fluxes_kernels_cuda.h"

namespace
{
    std::vector<std::string> get_variable_string(const std::string& var_name, std::vector<int> i_count, Netcdf_handle& nc, const int string_len)
    {
        const int total = std::accumulate(i_count.begin(), i_count.end(), 1, std::multiplies<>());
        i_count.push_back(string_len);
        const std::vector<char> chars = nc.get_variable<char>(var_name, i_count);
        std::vector<std::string> out;
        for (int n=0; n<total; ++n)
        {
            std::string s(chars.begin() + n*string_len, chars.begin() + (n+1)*string_len);
            const auto b = s.find_first_not_of(std::string(" \0", 2));
            const auto e = s.find_last_not_of(std::string(" \0", 2));
            out.push_back(b == std::string::npos ? std::string() : s.substr(b, e - b + 1));
        }
        return out;
    }

    Gas_optics_rrtmgp_gpu load_and_init_gas_optics(const Gas_concs_gpu& gas_concs, const std::string& coef_file)
    {
        Netcdf_file coef_nc(coef_file, Netcdf_mode::Read);

        const int n_temps = coef_nc.get_dimension_size("temperature");
        const int n_press = coef_nc.get_dimension_size("pressure");
        const int n_absorbers = coef_nc.get_dimension_size("absorber");
        const int n_char = coef_nc.get_dimension_size("string_len");
        const int n_minorabsorbers = coef_nc.get_dimension_size("minor_absorber");
        const int n_extabsorbers = coef_nc.get_dimension_size("absorber_ext");
        const int n_mixingfracs = coef_nc.get_dimension_size("mixing_fraction");
        const int n_layers = coef_nc.get_dimension_size("atmos_layer");
        const int n_bnds = coef_nc.get_dimension_size("bnd");
        const int n_gpts = coef_nc.get_dimension_size("gpt");
        const int n_pairs = coef_nc.get_dimension_size("pair");
        const int n_lower = coef_nc.get_dimension_size("minor_absorber_intervals_lower");
        const int n_upper = coef_nc.get_dimension_size("minor_absorber_intervals_upper");
        const int n_contributors_lower = coef_nc.get_dimension_size("contributors_lower");
        const int n_contributors_upper = coef_nc.get_dimension_size("contributors_upper");

        // NetCDF (C order) dimensions are the reverse of the Array (column-major) dimensions.
        Array<std::string,1> gas_names(get_variable_string("gas_names", {n_absorbers}, coef_nc, n_char), {n_absorbers});
        Array<int,3> key_species(coef_nc.get_variable<int>("key_species", {n_bnds, n_layers, 2}), {2, n_layers, n_bnds});
        Array<Float,2> band_lims(coef_nc.get_variable<Float>("bnd_limits_wavenumber", {n_bnds, 2}), {2, n_bnds});
        Array<int,2> band2gpt(coef_nc.get_variable<int>("bnd_limits_gpt", {n_bnds, 2}), {2, n_bnds});
        Array<Float,1> press_ref(coef_nc.get_variable<Float>("press_ref", {n_press}), {n_press});
        Array<Float,1> temp_ref(coef_nc.get_variable<Float>("temp_ref", {n_temps}), {n_temps});
        const Float temp_ref_p = coef_nc.get_variable<Float>("absorption_coefficient_ref_P");
        const Float temp_ref_t = coef_nc.get_variable<Float>("absorption_coefficient_ref_T");
        const Float press_ref_trop = coef_nc.get_variable<Float>("press_ref_trop");

        Array<Float,3> kminor_lower(coef_nc.get_variable<Float>("kminor_lower", {n_temps, n_mixingfracs, n_contributors_lower}),
                                    {n_contributors_lower, n_mixingfracs, n_temps});
        Array<Float,3> kminor_upper(coef_nc.get_variable<Float>("kminor_upper", {n_temps, n_mixingfracs, n_contributors_upper}),
                                    {n_contributors_upper, n_mixingfracs, n_temps});
        Array<std::string,1> gas_minor(get_variable_string("gas_minor", {n_minorabsorbers}, coef_nc, n_char), {n_minorabsorbers});
        Array<std::string,1> identifier_minor(get_variable_string("identifier_minor", {n_minorabsorbers}, coef_nc, n_char), {n_minorabsorbers});
        Array<std::string,1> minor_gases_lower(get_variable_string("minor_gases_lower", {n_lower}, coef_nc, n_char), {n_lower});
        Array<std::string,1> minor_gases_upper(get_variable_string("minor_gases_upper", {n_upper}, coef_nc, n_char), {n_upper});
        Array<int,2> minor_limits_gpt_lower(coef_nc.get_variable<int>("minor_limits_gpt_lower", {n_lower, n_pairs}), {n_pairs, n_lower});
        Array<int,2> minor_limits_gpt_upper(coef_nc.get_variable<int>("minor_limits_gpt_upper", {n_upper, n_pairs}), {n_pairs, n_upper});
        Array<Bool,1> minor_scales_with_density_lower(coef_nc.get_variable<Bool>("minor_scales_with_density_lower", {n_lower}), {n_lower});
        Array<Bool,1> minor_scales_with_density_upper(coef_nc.get_variable<Bool>("minor_scales_with_density_upper", {n_upper}), {n_upper});
        Array<Bool,1> scale_by_complement_lower(coef_nc.get_variable<Bool>("scale_by_complement_lower", {n_lower}), {n_lower});
        Array<Bool,1> scale_by_complement_upper(coef_nc.get_variable<Bool>("scale_by_complement_upper", {n_upper}), {n_upper});
        Array<std::string,1> scaling_gas_lower(get_variable_string("scaling_gas_lower", {n_lower}, coef_nc, n_char), {n_lower});
        Array<std::string,1> scaling_gas_upper(get_variable_string("scaling_gas_upper", {n_upper}, coef_nc, n_char), {n_upper});
        Array<int,1> kminor_start_lower(coef_nc.get_variable<int>("kminor_start_lower", {n_lower}), {n_lower});
        Array<int,1> kminor_start_upper(coef_nc.get_variable<int>("kminor_start_upper", {n_upper}), {n_upper});
        Array<Float,3> vmr_ref(coef_nc.get_variable<Float>("vmr_ref", {n_temps, n_extabsorbers, n_layers}), {n_layers, n_extabsorbers, n_temps});
        Array<Float,4> kmajor(coef_nc.get_variable<Float>("kmajor", {n_temps, n_press+1, n_mixingfracs, n_gpts}),
                              {n_gpts, n_mixingfracs, n_press+1, n_temps});

        Array<Float,3> rayl_lower, rayl_upper;
        if (coef_nc.variable_exists("rayl_lower"))
        {
            rayl_lower = Array<Float,3>(coef_nc.get_variable<Float>("rayl_lower", {n_temps, n_mixingfracs, n_gpts}), {n_gpts, n_mixingfracs, n_temps});
            rayl_upper = Array<Float,3>(coef_nc.get_variable<Float>("rayl_upper", {n_temps, n_mixingfracs, n_gpts}), {n_gpts, n_mixingfracs, n_temps});
        }

        if (coef_nc.variable_exists("totplnk"))
        {
            const int n_internal_sourcetemps = coef_nc.get_dimension_size("temperature_Planck");
            Array<Float,2> totplnk(coef_nc.get_variable<Float>("totplnk", {n_bnds, n_internal_sourcetemps}), {n_internal_sourcetemps, n_bnds});
            Array<Float,4> planck_frac(coef_nc.get_variable<Float>("plank_fraction", {n_temps, n_press+1, n_mixingfracs, n_gpts}),
                                       {n_gpts, n_mixingfracs, n_press+1, n_temps});
            return Gas_optics_rrtmgp_gpu(
                    gas_concs, gas_names, key_species, band2gpt, band_lims, press_ref, press_ref_trop, temp_ref, temp_ref_p, temp_ref_t,
                    vmr_ref, kmajor, kminor_lower, kminor_upper, gas_minor, identifier_minor, minor_gases_lower, minor_gases_upper,
                    minor_limits_gpt_lower, minor_limits_gpt_upper, minor_scales_with_density_lower, minor_scales_with_density_upper,
                    scaling_gas_lower, scaling_gas_upper, scale_by_complement_lower, scale_by_complement_upper,
                    kminor_start_lower, kminor_start_upper, totplnk, planck_frac, rayl_lower, rayl_upper);
        }
        else
        {
            Array<Float,1> solar_src_quiet(coef_nc.get_variable<Float>("solar_source_quiet", {n_gpts}), {n_gpts});
            Array<Float,1> solar_src_facular(coef_nc.get_variable<Float>("solar_source_facular", {n_gpts}), {n_gpts});
            Array<Float,1> solar_src_sunspot(coef_nc.get_variable<Float>("solar_source_sunspot", {n_gpts}), {n_gpts});
            const Float tsi = coef_nc.get_variable<Float>("tsi_default");
            const Float mg_index = coef_nc.get_variable<Float>("mg_default");
            const Float sb_index = coef_nc.get_variable<Float>("sb_default");
            return Gas_optics_rrtmgp_gpu(
                    gas_concs, gas_names, key_species, band2gpt, band_lims, press_ref, press_ref_trop, temp_ref, temp_ref_p, temp_ref_t,
                    vmr_ref, kmajor, kminor_lower, kminor_upper, gas_minor, identifier_minor, minor_gases_lower, minor_gases_upper,
                    minor_limits_gpt_lower, minor_limits_gpt_upper, minor_scales_with_density_lower, minor_scales_with_density_upper,
                    scaling_gas_lower, scaling_gas_upper, scale_by_complement_lower, scale_by_complement_upper,
                    kminor_start_lower, kminor_start_upper, solar_src_quiet, solar_src_facular, solar_src_sunspot,
                    tsi, mg_index, sb_index, rayl_lower, rayl_upper);
        }
    }

    Cloud_optics_gpu load_and_init_cloud_optics(const std::string& coef_file)
    {
        Netcdf_file coef_nc(coef_file, Netcdf_mode::Read);
        const int n_band = coef_nc.get_dimension_size("nband");
        const int n_rghice = coef_nc.get_dimension_size("nrghice");
        const int n_size_liq = coef_nc.get_dimension_size("nsize_liq");
        const int n_size_ice = coef_nc.get_dimension_size("nsize_ice");
        Array<Float,2> band_lims_wvn(coef_nc.get_variable<Float>("bnd_limits_wavenumber", {n_band, 2}), {2, n_band});
        const Float radliq_lwr = coef_nc.get_variable<Float>("radliq_lwr");
        const Float radliq_upr = coef_nc.get_variable<Float>("radliq_upr");
        const Float radliq_fac = coef_nc.get_variable<Float>("radliq_fac");
        const Float diamice_lwr = coef_nc.get_variable<Float>("diamice_lwr");
        const Float diamice_upr = coef_nc.get_variable<Float>("diamice_upr");
        const Float diamice_fac = coef_nc.get_variable<Float>("diamice_fac");
        Array<Float,2> lut_extliq(coef_nc.get_variable<Float>("lut_extliq", {n_band, n_size_liq}), {n_size_liq, n_band});
        Array<Float,2> lut_ssaliq(coef_nc.get_variable<Float>("lut_ssaliq", {n_band, n_size_liq}), {n_size_liq, n_band});
        Array<Float,2> lut_asyliq(coef_nc.get_variable<Float>("lut_asyliq", {n_band, n_size_liq}), {n_size_liq, n_band});
        Array<Float,3> lut_extice(coef_nc.get_variable<Float>("lut_extice", {n_rghice, n_band, n_size_ice}), {n_size_ice, n_band, n_rghice});
        Array<Float,3> lut_ssaice(coef_nc.get_variable<Float>("lut_ssaice", {n_rghice, n_band, n_size_ice}), {n_size_ice, n_band, n_rghice});
        Array<Float,3> lut_asyice(coef_nc.get_variable<Float>("lut_asyice", {n_rghice, n_band, n_size_ice}), {n_size_ice, n_band, n_rghice});
        return Cloud_optics_gpu(band_lims_wvn, radliq_lwr, radliq_upr, radliq_fac, diamice_lwr, diamice_upr, diamice_fac,
                                lut_extliq, lut_ssaliq, lut_asyliq, lut_extice, lut_ssaice, lut_asyice);
    }

    // /root/reference/src_test/Radiation_solver.cu:366-401. The file holds no band limits for these tables: the reference hands
    // over an all-zero (2, n_band) array, and so does this loader (only the band COUNT is used, by add_to).
    Aerosol_optics_gpu load_and_init_aerosol_optics(const std::string& coef_file)
    {
        Netcdf_file coef_nc(coef_file, Netcdf_mode::Read);
        const int n_band = coef_nc.get_dimension_size("band_sw");
        const int n_hum = coef_nc.get_dimension_size("relative_humidity");
        const int n_philic = coef_nc.get_dimension_size("hydrophilic");
        const int n_phobic = coef_nc.get_dimension_size("hydrophobic");
        Array<Float,2> band_lims_wvn({2, n_band});
        Array<Float,2> mext_phobic(coef_nc.get_variable<Float>("mass_ext_sw_hydrophobic", {n_phobic, n_band}), {n_band, n_phobic});
        Array<Float,2> ssa_phobic(coef_nc.get_variable<Float>("ssa_sw_hydrophobic", {n_phobic, n_band}), {n_band, n_phobic});
        Array<Float,2> g_phobic(coef_nc.get_variable<Float>("asymmetry_sw_hydrophobic", {n_phobic, n_band}), {n_band, n_phobic});
        Array<Float,3> mext_philic(coef_nc.get_variable<Float>("mass_ext_sw_hydrophilic", {n_philic, n_hum, n_band}), {n_band, n_hum, n_philic});
        Array<Float,3> ssa_philic(coef_nc.get_variable<Float>("ssa_sw_hydrophilic", {n_philic, n_hum, n_band}), {n_band, n_hum, n_philic});
        Array<Float,3> g_philic(coef_nc.get_variable<Float>("asymmetry_sw_hydrophilic", {n_philic, n_hum, n_band}), {n_band, n_hum, n_philic});
        Array<Float,1> rh_upper(coef_nc.get_variable<Float>("relative_humidity2", {n_hum}), {n_hum});
        return Aerosol_optics_gpu(band_lims_wvn, rh_upper, mext_phobic, ssa_phobic, g_phobic, mext_philic, ssa_philic, g_philic);
    }

    // contiguous column blocks {1-based start, size}
    std::vector<std::pair<int,int>> column_blocks(const int n_col, const int n_col_block)
    {
        std::vector<std::pair<int,int>> b;
        for (int s=1; s<=n_col; s+=n_col_block) b.emplace_back(s, std::min(n_col_block, n_col - s + 1));
        return b;
    }

    // ---- column order of a solve (include_test/Radiation_solver.h: set_column_sorting / set_column_padding) ----
    // A gather index over the caller's columns: sorted by surface pressure and / or padded to a multiple of 16 by repeating the last one.
    struct Column_order
    {
        int n_col = 0, n_out = 0;
        Array_gpu<int,1> perm;
        bool active() const { return n_out > 0; }
        // (col, n2) -> (n_out, n2); an absent (empty) array stays absent
        Array_gpu<Float,2> in2(const Array_gpu<Float,2>& a) const
        {
            if (a.size() == 0) return Array_gpu<Float,2>();
            Array_gpu<Float,2> o({n_out, a.dim(2)});
            RRX_CALL(rrx_gather_cols, n_out, (unsigned long long)a.dim(2), perm.ptr(), n_col, a.ptr(), o.ptr());
            return o;
        }
        Array_gpu<Float,1> in1(const Array_gpu<Float,1>& a) const
        {
            if (a.size() == 0) return Array_gpu<Float,1>();
            Array_gpu<Float,1> o({n_out});
            RRX_CALL(rrx_gather_cols, n_out, 1ull, perm.ptr(), n_col, a.ptr(), o.ptr());
            return o;
        }
        // (n1, col) -> (n1, n_out)
        Array_gpu<Float,2> in_last(const Array_gpu<Float,2>& a) const
        {
            if (a.size() == 0) return Array_gpu<Float,2>();
            Array_gpu<Float,2> o({a.dim(1), n_out});
            RRX_CALL(rrx_gather_lastdim, a.dim(1), n_out, perm.ptr(), a.ptr(), o.ptr());
            return o;
        }
        // results of the reordered solve back into the caller's arrays (first n_col entries of perm: the permutation proper)
        template<int N> void out(Array_gpu<Float,N>& dst, const Array_gpu<Float,N>& src) const
        {
            if (src.size() == 0) return;
            std::array<int,N> d; d[0] = n_col; unsigned long long rest = 1;
            for (int i=1; i<N; ++i) { d[i] = src.dim(i+1); rest *= (unsigned long long)src.dim(i+1); }
            if (dst.size() == 0) dst.set_dims(d);
            RRX_CALL(rrx_scatter_cols, n_col, rest, perm.ptr(), n_out, src.ptr(), n_col, dst.ptr());
        }
    };

    // Should this solve reorder its columns, and how? `sort_decided` caches the automatic decision of the solver object.
    Column_order column_order(const int mode, int& sort_decided, const bool pad, const Array_gpu<Float,2>& p_lev, const Bool top_at_1)
    {
        Column_order co;
        const int n_col = p_lev.dim(1), n_lev = p_lev.dim(2);
        const Float* p_sfc = p_lev.ptr() + size_t(top_at_1 ? n_lev-1 : 0)*n_col;
        bool sort = mode == 1;
        if (mode < 0 && n_col >= 256)
        {
            if (sort_decided < 0)
            {
                Array_gpu<int,1> flag({1});
                RRX_CALL(rrx_column_spread, n_col, p_sfc, 256, Float(0.2), flag.ptr());
                int h = 0;
                rrx_host::check(rrx_memcpy_d2h_stream(&h, flag.ptr(), sizeof(int), rrx_host::current_stream()));     // (synchronises: once per solver)
                sort_decided = h ? 1 : 0;
            }
            sort = sort_decided == 1;
        }
        const int n_pad = (pad && n_col > 16 && n_col % 16 != 0) ? 16 - n_col % 16 : 0;
        if (!sort && n_pad == 0) return co;
        co.n_col = n_col; co.n_out = n_col + n_pad;
        co.perm.set_dims({co.n_out});
        if (sort) RRX_CALL(rrx_sort_columns, n_col, p_sfc, n_pad, co.perm.ptr());
        else rrx_host::check(rrx_identity_columns(n_col, n_pad, co.perm.ptr(), rrx_host::current_stream()));
        return co;
    }
}


void compute_heating_rate(const Array_gpu<Float,2>& flux_net, const Array_gpu<Float,2>& p_lev, Array_gpu<Float,2>& heating_rate)
{
    const int n_col = flux_net.dim(1), n_lev = flux_net.dim(2);
    if (p_lev.dim(1) != n_col || p_lev.dim(2) != n_lev) throw std::runtime_error("compute_heating_rate: flux and pressure shapes differ");
    if (heating_rate.size() == 0) heating_rate.set_dims({n_col, n_lev-1});
    RRX_CALL(rrx_heating_rate, n_col, n_lev-1, Float(9.80665/1004.64), flux_net.ptr(), p_lev.ptr(), heating_rate.ptr());
}


// -------------------------------------------------------------------------------------------- longwave
struct Radiation_solver_longwave::Workspace
{
    int n_col = 0, n_lay = 0;
    bool broadband = false;
    std::unique_ptr<Optical_props_arry_gpu> optical_props;
    std::unique_ptr<Optical_props_1scl_gpu> cloud_optical_props;
    std::unique_ptr<Source_func_lw_gpu> sources;
    Array_gpu<Float,3> gpt_flux_up, gpt_flux_dn;
};

Radiation_solver_longwave::Radiation_solver_longwave(
        const Gas_concs_gpu& gas_concs, const std::string& file_name_gas, const std::string& file_name_cloud)
{
    this->kdist_gpu = std::make_unique<Gas_optics_rrtmgp_gpu>(load_and_init_gas_optics(gas_concs, file_name_gas));
    if (!file_name_cloud.empty())
        this->cloud_optics_gpu = std::make_unique<Cloud_optics_gpu>(load_and_init_cloud_optics(file_name_cloud));
}

void Radiation_solver_longwave::solve_gpu(
        const bool switch_fluxes,
        const bool switch_cloud_optics,
        const bool switch_output_optical,
        const bool switch_output_bnd_fluxes,
        const Gas_concs_gpu& gas_concs,
        const Array_gpu<Float,2>& p_lay, const Array_gpu<Float,2>& p_lev,
        const Array_gpu<Float,2>& t_lay, const Array_gpu<Float,2>& t_lev,
        const Array_gpu<Float,2>& col_dry,
        const Array_gpu<Float,1>& t_sfc, const Array_gpu<Float,2>& emis_sfc,
        const Array_gpu<Float,2>& lwp, const Array_gpu<Float,2>& iwp,
        const Array_gpu<Float,2>& rel, const Array_gpu<Float,2>& dei,
        Array_gpu<Float,3>& tau, Array_gpu<Float,3>& lay_source,
        Array_gpu<Float,3>& lev_source, Array_gpu<Float,2>& sfc_source,
        Array_gpu<Float,2>& lw_flux_up, Array_gpu<Float,2>& lw_flux_dn, Array_gpu<Float,2>& lw_flux_net,
        Array_gpu<Float,3>& lw_bnd_flux_up, Array_gpu<Float,3>& lw_bnd_flux_dn, Array_gpu<Float,3>& lw_bnd_flux_net)
{
    const int n_col = p_lay.dim(1);
    const int n_lay = p_lay.dim(2);
    const int n_lev = p_lev.dim(2);
    const int n_gpt = this->kdist_gpu->get_ngpt();
    const int n_bnd = this->kdist_gpu->get_nband();
    const Bool top_at_1 = (vertical_ordering < 0) ? Bool(p_lay({1, 1}) < p_lay({1, n_lay})) : Bool(vertical_ordering == 1);
    if (switch_cloud_optics && !cloud_optics_gpu) throw std::runtime_error("cloud optics requested but no cloud coefficients loaded");
    const bool broadband = broadband_solvers && !switch_output_bnd_fluxes;

    // columns in another order / on a padded count: gather the inputs, solve, scatter the fluxes back (see the header)
    if (!reordered_call && !switch_output_optical && switch_fluxes)
    {
        const Column_order co = column_order(column_sorting, sort_decided, column_padding, p_lev, top_at_1);
        if (co.active())
        {
            const Gas_concs_gpu gases = gas_concs.gathered(co.perm, n_col, co.n_out);
            Array_gpu<Float,3> no3a, no3b, no3c; Array_gpu<Float,2> no2;
            Array_gpu<Float,2> up, dn, net; Array_gpu<Float,3> bup, bdn, bnet;
            up.set_dims({co.n_out, n_lev}); dn.set_dims({co.n_out, n_lev}); net.set_dims({co.n_out, n_lev});
            if (switch_output_bnd_fluxes) { bup.set_dims({co.n_out, n_lev, n_bnd}); bdn.set_dims({co.n_out, n_lev, n_bnd}); bnet.set_dims({co.n_out, n_lev, n_bnd}); }
            struct Guard { bool& f; Guard(bool& f_) : f(f_) { f = true; } ~Guard() { f = false; } } guard(reordered_call);
            this->solve_gpu(switch_fluxes, switch_cloud_optics, switch_output_optical, switch_output_bnd_fluxes, gases,
                            co.in2(p_lay), co.in2(p_lev), co.in2(t_lay), co.in2(t_lev), co.in2(col_dry), co.in1(t_sfc), co.in_last(emis_sfc),
                            co.in2(lwp), co.in2(iwp), co.in2(rel), co.in2(dei), no3a, no3b, no3c, no2, up, dn, net, bup, bdn, bnet);
            co.out(lw_flux_up, up); co.out(lw_flux_dn, dn); co.out(lw_flux_net, net);
            if (switch_output_bnd_fluxes) { co.out(lw_bnd_flux_up, bup); co.out(lw_bnd_flux_dn, bdn); co.out(lw_bnd_flux_net, bnet); }
            return;
        }
    }

    auto prepare = [&](std::shared_ptr<Workspace>& ws, const int n)
    {
        if (!ws || ws->n_col != n || ws->n_lay != n_lay || ws->broadband != broadband)
        {
            ws = std::make_shared<Workspace>();
            ws->n_col = n; ws->n_lay = n_lay; ws->broadband = broadband;
            ws->optical_props = std::make_unique<Optical_props_1scl_gpu>(n, n_lay, *kdist_gpu);
            ws->sources = std::make_unique<Source_func_lw_gpu>(n, n_lay, *kdist_gpu);
            // broadband solver: Planck fractions instead of the two source arrays (they are materialised on demand, e.g. for
            // --output-optical)
            ws->sources->enable_planck_lite(broadband);
            ws->gpt_flux_up.set_dims({n, n_lev, broadband ? 1 : n_gpt});
            ws->gpt_flux_dn.set_dims({n, n_lev, broadband ? 1 : n_gpt});
        }
        if (switch_cloud_optics && !ws->cloud_optical_props)
            ws->cloud_optical_props = std::make_unique<Optical_props_1scl_gpu>(n, n_lay, *cloud_optics_gpu);
    };

    for (const auto& blk : column_blocks(n_col, std::max(1, n_col_block)))
    {
        const int col_s = blk.first, n_in = blk.second, col_e = col_s + n_in - 1;
        std::shared_ptr<Workspace>& wsp = (n_in == std::min(n_col_block, n_col)) ? ws_block : ws_residual;
        prepare(wsp, n_in);
        Workspace& ws = *wsp;
        const bool whole = (n_in == n_col);      // a single block needs no gather of the inputs

        // (a single block takes the caller's gases as they are: no device copies of the mixing-ratio fields)
        std::unique_ptr<Gas_concs_gpu> gas_concs_copy;
        if (!whole) gas_concs_copy = std::make_unique<Gas_concs_gpu>(gas_concs, col_s, n_in);
        const Gas_concs_gpu& gas_concs_subset = whole ? gas_concs : *gas_concs_copy;
        auto sub2 = [&](const Array_gpu<Float,2>& a, const int n2) { return whole ? Array_gpu<Float,2>(const_cast<Float*>(a.ptr()), {n_in, n2})
                                                                                  : a.subset({{ {col_s, col_e}, {1, n2} }}); };
        Array_gpu<Float,2> p_lay_s = sub2(p_lay, n_lay), t_lay_s = sub2(t_lay, n_lay);
        Array_gpu<Float,2> p_lev_s = sub2(p_lev, n_lev), t_lev_s = sub2(t_lev, n_lev);
        Array_gpu<Float,1> t_sfc_s = whole ? Array_gpu<Float,1>(const_cast<Float*>(t_sfc.ptr()), {n_in}) : t_sfc.subset({{ {col_s, col_e} }});

        Array_gpu<Float,2> col_dry_s({n_in, n_lay});
        if (col_dry.size() == 0)
            Gas_optics_rrtmgp_gpu::get_col_dry(col_dry_s, gas_concs_subset.get_vmr("h2o"), p_lev_s);
        else
            col_dry_s = sub2(col_dry, n_lay);

        // (cloud optics first: its by-band optical depth is added inside gas_optics where tau is stored -- the add_to() of
        //  Radiation_solver.cu:508-511 folded into the producer)
        if (switch_cloud_optics)
            cloud_optics_gpu->cloud_optics(sub2(lwp, n_lay), sub2(iwp, n_lay), sub2(rel, n_lay), sub2(dei, n_lay), *ws.cloud_optical_props);
        kdist_gpu->gas_optics(p_lay_s, p_lev_s, t_lay_s, t_sfc_s, gas_concs_subset, ws.optical_props, *ws.sources, col_dry_s, t_lev_s,
                              switch_cloud_optics ? ws.cloud_optical_props.get() : nullptr);

        if (switch_output_optical)
        {
            // (the reference scatters lev_source with n_lay rows, Radiation_solver.cu:520-523, which drops the last level
            //  and misplaces the others; lev_source has n_lev rows)
            Float* full_lay[2] = {tau.ptr(), lay_source.ptr()};
            const Float* sub_lay[2] = {ws.optical_props->get_tau().ptr(), ws.sources->get_lay_source().ptr()};
            Subset_kernels_cuda::scatter_(n_col, n_lay, n_gpt, n_in, col_s, 2, full_lay, sub_lay);
            Float* full_lev[1] = {lev_source.ptr()};
            const Float* sub_lev[1] = {ws.sources->get_lev_source().ptr()};
            Subset_kernels_cuda::scatter_(n_col, n_lev, n_gpt, n_in, col_s, 1, full_lev, sub_lev);
            Subset_kernels_cuda::get_from_subset(n_col, n_gpt, n_in, col_s, sfc_source.ptr(), ws.sources->get_sfc_source().ptr());
        }
        if (!switch_fluxes)
            continue;

        constexpr int n_ang = 1;
        Array_gpu<Float,2> emis_s = whole ? Array_gpu<Float,2>(const_cast<Float*>(emis_sfc.ptr()), {n_bnd, n_in}) : emis_sfc.subset({{ {1, n_bnd}, {col_s, col_e} }});
        if (whole && broadband && !switch_output_bnd_fluxes)
        {
            // one block in broadband mode: the solver writes the caller's flux arrays, the net flux follows in place (no block
            // workspace, no copies: Fluxes_broadband_gpu::reduce + get_from_subset of the general path are 7 passes over the fluxes)
            if (lw_flux_up.size() == 0) lw_flux_up.set_dims({n_col, n_lev});
            if (lw_flux_dn.size() == 0) lw_flux_dn.set_dims({n_col, n_lev});
            if (lw_flux_net.size() == 0) lw_flux_net.set_dims({n_col, n_lev});
            Array_gpu<Float,3> up3(lw_flux_up.ptr(), {n_col, n_lev, 1}), dn3(lw_flux_dn.ptr(), {n_col, n_lev, 1});
            rte_lw.rte_lw(ws.optical_props, top_at_1, *ws.sources, emis_s, Array_gpu<Float,2>(), up3, dn3, n_ang);
            Fluxes_kernels_cuda::net_broadband_precalc(n_col, n_lev, lw_flux_dn.ptr(), lw_flux_up.ptr(), lw_flux_net.ptr());
            continue;
        }
        rte_lw.rte_lw(ws.optical_props, top_at_1, *ws.sources, emis_s, Array_gpu<Float,2>(), ws.gpt_flux_up, ws.gpt_flux_dn, n_ang);

        Fluxes_broadband_gpu fluxes(n_in, n_lev);
        fluxes.reduce(ws.gpt_flux_up, ws.gpt_flux_dn, ws.optical_props, top_at_1);
        Subset_kernels_cuda::get_from_subset(n_col, n_lev, n_in, col_s, lw_flux_up.ptr(), lw_flux_dn.ptr(), lw_flux_net.ptr(),
                fluxes.get_flux_up().ptr(), fluxes.get_flux_dn().ptr(), fluxes.get_flux_net().ptr());

        if (switch_output_bnd_fluxes)
        {
            Fluxes_byband_gpu bnd_fluxes(n_in, n_lev, n_bnd);
            bnd_fluxes.reduce(ws.gpt_flux_up, ws.gpt_flux_dn, ws.optical_props, top_at_1);
            Subset_kernels_cuda::get_from_subset(n_col, n_lev, n_bnd, n_in, col_s, lw_bnd_flux_up.ptr(), lw_bnd_flux_dn.ptr(), lw_bnd_flux_net.ptr(),
                    bnd_fluxes.get_bnd_flux_up().ptr(), bnd_fluxes.get_bnd_flux_dn().ptr(), bnd_fluxes.get_bnd_flux_net().ptr());
        }
    }
}


// -------------------------------------------------------------------------------------------- shortwave
struct Radiation_solver_shortwave::Workspace
{
    int n_col = 0, n_lay = 0;
    bool broadband = false;
    std::unique_ptr<Optical_props_arry_gpu> optical_props;
    std::unique_ptr<Optical_props_2str_gpu> cloud_optical_props, aerosol_optical_props;
    Array_gpu<Float,3> gpt_flux_up, gpt_flux_dn, gpt_flux_dn_dir;
};

Radiation_solver_shortwave::Radiation_solver_shortwave(
        const Gas_concs_gpu& gas_concs,
        const bool switch_cloud_optics,
        const bool switch_aerosol_optics,
        const std::string& file_name_gas,
        const std::string& file_name_cloud,
        const std::string& file_name_aerosol)
{
    this->kdist_gpu = std::make_unique<Gas_optics_rrtmgp_gpu>(load_and_init_gas_optics(gas_concs, file_name_gas));
    if (switch_cloud_optics)
        this->cloud_optics_gpu = std::make_unique<Cloud_optics_gpu>(load_and_init_cloud_optics(file_name_cloud));
    if (switch_aerosol_optics)
    {
        this->aerosol_optics_gpu = std::make_unique<Aerosol_optics_gpu>(load_and_init_aerosol_optics(file_name_aerosol));
        if (this->aerosol_optics_gpu->get_nband() != this->kdist_gpu->get_nband())
            throw std::runtime_error("aerosol optics tables and the shortwave k-distribution disagree in the number of bands");
    }
}

void Radiation_solver_shortwave::solve_gpu(
        const bool switch_fluxes,
        const bool switch_cloud_optics,
        const bool switch_aerosol_optics,
        const bool switch_output_optical,
        const bool switch_output_bnd_fluxes,
        const bool switch_delta_cloud,
        const bool switch_delta_aerosol,
        const Gas_concs_gpu& gas_concs,
        const Array_gpu<Float,2>& p_lay, const Array_gpu<Float,2>& p_lev,
        const Array_gpu<Float,2>& t_lay, const Array_gpu<Float,2>& t_lev,
        const Array_gpu<Float,2>& col_dry,
        const Array_gpu<Float,2>& sfc_alb_dir, const Array_gpu<Float,2>& sfc_alb_dif,
        const Array_gpu<Float,1>& tsi_scaling, const Array_gpu<Float,1>& mu0,
        const Array_gpu<Float,2>& lwp, const Array_gpu<Float,2>& iwp,
        const Array_gpu<Float,2>& rel, const Array_gpu<Float,2>& dei,
        const Array_gpu<Float,2>& rh,
        const Aerosol_concs_gpu& aerosol_concs,
        Array_gpu<Float,3>& tau, Array_gpu<Float,3>& ssa, Array_gpu<Float,3>& g,
        Array_gpu<Float,2>& toa_src,
        Array_gpu<Float,2>& sw_flux_up, Array_gpu<Float,2>& sw_flux_dn,
        Array_gpu<Float,2>& sw_flux_dn_dir, Array_gpu<Float,2>& sw_flux_net,
        Array_gpu<Float,3>& sw_bnd_flux_up, Array_gpu<Float,3>& sw_bnd_flux_dn,
        Array_gpu<Float,3>& sw_bnd_flux_dn_dir, Array_gpu<Float,3>& sw_bnd_flux_net)
{
    (void)t_lev;
    const int n_col = p_lay.dim(1);
    const int n_lay = p_lay.dim(2);
    const int n_lev = p_lev.dim(2);
    const int n_gpt = this->kdist_gpu->get_ngpt();
    const int n_bnd = this->kdist_gpu->get_nband();
    const Bool top_at_1 = (vertical_ordering < 0) ? Bool(p_lay({1, 1}) < p_lay({1, n_lay})) : Bool(vertical_ordering == 1);
    if (switch_cloud_optics && !cloud_optics_gpu) throw std::runtime_error("cloud optics requested but no cloud coefficients loaded");
    if (switch_aerosol_optics && !aerosol_optics_gpu) throw std::runtime_error("aerosol optics requested but no aerosol coefficients loaded");
    const bool broadband = broadband_solvers && !switch_output_bnd_fluxes;

    // columns in another order / on a padded count: gather the inputs, solve, scatter the fluxes back (see the header)
    if (!reordered_call && !switch_output_optical && switch_fluxes)
    {
        const Column_order co = column_order(column_sorting, sort_decided, column_padding, p_lev, top_at_1);
        if (co.active())
        {
            const Gas_concs_gpu gases = gas_concs.gathered(co.perm, n_col, co.n_out);
            Aerosol_concs_gpu aerosols = aerosol_concs.gathered(co.perm, n_col, co.n_out);
            Array_gpu<Float,3> no3a, no3b, no3c; Array_gpu<Float,2> no2;
            Array_gpu<Float,2> up, dn, dir, net; Array_gpu<Float,3> bup, bdn, bdir, bnet;
            up.set_dims({co.n_out, n_lev}); dn.set_dims({co.n_out, n_lev}); dir.set_dims({co.n_out, n_lev}); net.set_dims({co.n_out, n_lev});
            if (switch_output_bnd_fluxes)
            { bup.set_dims({co.n_out, n_lev, n_bnd}); bdn.set_dims({co.n_out, n_lev, n_bnd}); bdir.set_dims({co.n_out, n_lev, n_bnd}); bnet.set_dims({co.n_out, n_lev, n_bnd}); }
            struct Guard { bool& f; Guard(bool& f_) : f(f_) { f = true; } ~Guard() { f = false; } } guard(reordered_call);
            this->solve_gpu(switch_fluxes, switch_cloud_optics, switch_aerosol_optics, switch_output_optical, switch_output_bnd_fluxes,
                            switch_delta_cloud, switch_delta_aerosol, gases,
                            co.in2(p_lay), co.in2(p_lev), co.in2(t_lay), co.in2(t_lev), co.in2(col_dry),
                            co.in_last(sfc_alb_dir), co.in_last(sfc_alb_dif), co.in1(tsi_scaling), co.in1(mu0),
                            co.in2(lwp), co.in2(iwp), co.in2(rel), co.in2(dei), co.in2(rh), aerosols,
                            no3a, no3b, no3c, no2, up, dn, dir, net, bup, bdn, bdir, bnet);
            co.out(sw_flux_up, up); co.out(sw_flux_dn, dn); co.out(sw_flux_dn_dir, dir); co.out(sw_flux_net, net);
            if (switch_output_bnd_fluxes)
            { co.out(sw_bnd_flux_up, bup); co.out(sw_bnd_flux_dn, bdn); co.out(sw_bnd_flux_dn_dir, bdir); co.out(sw_bnd_flux_net, bnet); }
            return;
        }
    }

    auto prepare = [&](std::shared_ptr<Workspace>& ws, const int n)
    {
        if (!ws || ws->n_col != n || ws->n_lay != n_lay || ws->broadband != broadband)
        {
            ws = std::make_shared<Workspace>();
            ws->n_col = n; ws->n_lay = n_lay; ws->broadband = broadband;
            ws->optical_props = std::make_unique<Optical_props_2str_gpu>(n, n_lay, *kdist_gpu);
            const int ng = broadband ? 1 : n_gpt;
            ws->gpt_flux_up.set_dims({n, n_lev, ng}); ws->gpt_flux_dn.set_dims({n, n_lev, ng}); ws->gpt_flux_dn_dir.set_dims({n, n_lev, ng});
        }
        if (switch_cloud_optics && !ws->cloud_optical_props)
            ws->cloud_optical_props = std::make_unique<Optical_props_2str_gpu>(n, n_lay, *cloud_optics_gpu);
        if (switch_aerosol_optics && !ws->aerosol_optical_props)
            ws->aerosol_optical_props = std::make_unique<Optical_props_2str_gpu>(n, n_lay, *aerosol_optics_gpu);
    };

    for (const auto& blk : column_blocks(n_col, std::max(1, n_col_block)))
    {
        const int col_s = blk.first, n_in = blk.second, col_e = col_s + n_in - 1;
        std::shared_ptr<Workspace>& wsp = (n_in == std::min(n_col_block, n_col)) ? ws_block : ws_residual;
        prepare(wsp, n_in);
        Workspace& ws = *wsp;
        const bool whole = (n_in == n_col);

        // (a single block takes the caller's gases as they are: no device copies of the mixing-ratio fields)
        std::unique_ptr<Gas_concs_gpu> gas_concs_copy;
        if (!whole) gas_concs_copy = std::make_unique<Gas_concs_gpu>(gas_concs, col_s, n_in);
        const Gas_concs_gpu& gas_concs_subset = whole ? gas_concs : *gas_concs_copy;
        auto sub2 = [&](const Array_gpu<Float,2>& a, const int n2) { return whole ? Array_gpu<Float,2>(const_cast<Float*>(a.ptr()), {n_in, n2})
                                                                                  : a.subset({{ {col_s, col_e}, {1, n2} }}); };
        auto sub1 = [&](const Array_gpu<Float,1>& a) { return whole ? Array_gpu<Float,1>(const_cast<Float*>(a.ptr()), {n_in}) : a.subset({{ {col_s, col_e} }}); };
        Array_gpu<Float,2> p_lay_s = sub2(p_lay, n_lay), t_lay_s = sub2(t_lay, n_lay), p_lev_s = sub2(p_lev, n_lev);

        Array_gpu<Float,2> col_dry_s({n_in, n_lay});
        if (col_dry.size() == 0)
            Gas_optics_rrtmgp_gpu::get_col_dry(col_dry_s, gas_concs_subset.get_vmr("h2o"), p_lev_s);
        else
            col_dry_s = sub2(col_dry, n_lay);

        // (cloud optics first: gas and cloud properties are combined inside gas_optics where the g-point arrays are stored --
        //  the add_to() of Radiation_solver.cu:788-791 folded into the producer)
        if (switch_cloud_optics)
        {
            // (delta_scale() of Radiation_solver.cu:785 rides along in the same kernel)
            cloud_optics_gpu->cloud_optics(sub2(lwp, n_lay), sub2(iwp, n_lay), sub2(rel, n_lay), sub2(dei, n_lay), *ws.cloud_optical_props,
                                           switch_delta_cloud);
        }
        Array_gpu<Float,2> toa_src_s({n_in, n_gpt});
        kdist_gpu->gas_optics(p_lay_s, p_lev_s, t_lay_s, gas_concs_subset, ws.optical_props, toa_src_s, col_dry_s,
                              switch_cloud_optics ? ws.cloud_optical_props.get() : nullptr);
        Array_gpu<Float,1> tsi_s = sub1(tsi_scaling);
        RRX_CALL(rrx_scaling_to_subset, n_in, n_gpt, toa_src_s.ptr(), tsi_s.ptr());

        if (switch_aerosol_optics)
        {
            // the block's own columns (the reference subsets (1, n_col) here, Radiation_solver.cu:796, which is only right for
            // a single block)
            Aerosol_concs_gpu aerosol_concs_subset(aerosol_concs, col_s, n_in);
            aerosol_optics_gpu->aerosol_optics(aerosol_concs_subset, sub2(rh, n_lay), p_lev_s, *ws.aerosol_optical_props);
            if (switch_delta_aerosol)
                ws.aerosol_optical_props->delta_scale();
            add_to(dynamic_cast<Optical_props_2str_gpu&>(*ws.optical_props), *ws.aerosol_optical_props);
        }

        if (switch_output_optical)
        {
            Subset_kernels_cuda::get_from_subset(n_col, n_lay, n_gpt, n_in, col_s, tau.ptr(), ssa.ptr(), g.ptr(),
                    ws.optical_props->get_tau().ptr(), ws.optical_props->get_ssa().ptr(), ws.optical_props->get_g().ptr());
            Subset_kernels_cuda::get_from_subset(n_col, n_gpt, n_in, col_s, toa_src.ptr(), toa_src_s.ptr());
        }
        if (!switch_fluxes)
            continue;

        auto sub_last = [&](const Array_gpu<Float,2>& a) { return whole ? Array_gpu<Float,2>(const_cast<Float*>(a.ptr()), {n_bnd, n_in})
                                                                          : a.subset({{ {1, n_bnd}, {col_s, col_e} }}); };
        if (whole && broadband && !switch_output_bnd_fluxes)
        {
            // one block in broadband mode: the solver writes the caller's flux arrays, the net flux follows in place
            for (Array_gpu<Float,2>* a : {&sw_flux_up, &sw_flux_dn, &sw_flux_dn_dir, &sw_flux_net}) if (a->size() == 0) a->set_dims({n_col, n_lev});
            Array_gpu<Float,3> up3(sw_flux_up.ptr(), {n_col, n_lev, 1}), dn3(sw_flux_dn.ptr(), {n_col, n_lev, 1}), dir3(sw_flux_dn_dir.ptr(), {n_col, n_lev, 1});
            rte_sw.rte_sw(ws.optical_props, top_at_1, sub1(mu0), toa_src_s, sub_last(sfc_alb_dir), sub_last(sfc_alb_dif), Array_gpu<Float,2>(), up3, dn3, dir3);
            Fluxes_kernels_cuda::net_broadband_precalc(n_col, n_lev, sw_flux_dn.ptr(), sw_flux_up.ptr(), sw_flux_net.ptr());
            continue;
        }
        rte_sw.rte_sw(ws.optical_props, top_at_1, sub1(mu0), toa_src_s, sub_last(sfc_alb_dir), sub_last(sfc_alb_dif),
                Array_gpu<Float,2>(), ws.gpt_flux_up, ws.gpt_flux_dn, ws.gpt_flux_dn_dir);

        Fluxes_broadband_gpu fluxes(n_in, n_lev);
        fluxes.reduce(ws.gpt_flux_up, ws.gpt_flux_dn, ws.gpt_flux_dn_dir, ws.optical_props, top_at_1);
        Subset_kernels_cuda::get_from_subset(n_col, n_lev, n_in, col_s,
                sw_flux_up.ptr(), sw_flux_dn.ptr(), sw_flux_dn_dir.ptr(), sw_flux_net.ptr(),
                fluxes.get_flux_up().ptr(), fluxes.get_flux_dn().ptr(), fluxes.get_flux_dn_dir().ptr(), fluxes.get_flux_net().ptr());

        if (switch_output_bnd_fluxes)
        {
            Fluxes_byband_gpu bnd_fluxes(n_in, n_lev, n_bnd);
            bnd_fluxes.reduce(ws.gpt_flux_up, ws.gpt_flux_dn, ws.gpt_flux_dn_dir, ws.optical_props, top_at_1);
            Subset_kernels_cuda::get_from_subset(n_col, n_lev, n_bnd, n_in, col_s,
                    sw_bnd_flux_up.ptr(), sw_bnd_flux_dn.ptr(), sw_bnd_flux_dn_dir.ptr(), sw_bnd_flux_net.ptr(),
                    bnd_fluxes.get_bnd_flux_up().ptr(), bnd_fluxes.get_bnd_flux_dn().ptr(),
                    bnd_fluxes.get_bnd_flux_dn_dir().ptr(), bnd_fluxes.get_bnd_flux_net().ptr());
        }
    }
}
