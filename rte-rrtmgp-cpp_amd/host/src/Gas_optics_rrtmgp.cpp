// Gas_optics_rrtmgp_gpu: host-side bookkeeping of the k-distribution (what /root/reference/src/Gas_optics_rrtmgp.cpp:539-742
// and src_cuda/Gas_optics_rrtmgp.cu:427-803 do at construction) and the per-call launch sequence of
// src_cuda/Gas_optics_rrtmgp.cu:907-1201 on top of the C ABI of librrx_hip.so.
#include <algorithm>
#include <cmath>
#include <numeric>
#include "Gas_optics_rrtmgp.h"
#include "gas_optics_rrtmgp_kernels_cuda.h"

namespace
{
    std::string trimmed(const std::string& s)
    {
        const auto b = s.find_first_not_of(" \t\n\r\0", 0, 5);
        if (b == std::string::npos) return "";
        const auto e = s.find_last_not_of(" \t\n\r\0", std::string::npos, 5);
        return s.substr(b, e - b + 1);
    }

    // 1-based position of `value` in `names`, -1 if absent
    int find_index(const Array<std::string,1>& names, const std::string& value)
    {
        for (int i=1; i<=names.dim(1); ++i)
            if (names({i}) == value) return i;
        return -1;
    }

    // Minor-gas tables of one regime reduced to the gases the host model provides, with the absorption
    // coefficients reordered from the file's (ncontrib, neta, ntemp) to the kernels' (ntemp, neta, ncontrib).
    struct Minor_tables
    {
        Array<Float,3> kminor;
        Array<std::string,1> gases, scaling_gas;
        Array<int,2> limits_gpt;
        Array<Bool,1> scales_with_density, scale_by_complement;
        Array<int,1> kminor_start;
    };

    Minor_tables reduce_minor(
            const Gas_concs_gpu& available, const Array<std::string,1>& gas_minor, const Array<std::string,1>& identifier_minor,
            const Array<Float,3>& kminor, const Array<std::string,1>& minor_gases, const Array<int,2>& limits_gpt,
            const Array<Bool,1>& scales_with_density, const Array<std::string,1>& scaling_gas,
            const Array<Bool,1>& scale_by_complement, const Array<int,1>& kminor_start)
    {
        const int nm = minor_gases.dim(1);
        std::vector<int> keep;
        int ncontrib = 0;
        for (int i=1; i<=nm; ++i)
        {
            const int idx = find_index(identifier_minor, minor_gases({i}));
            if (idx < 0) throw std::runtime_error("Gas optics: unknown minor-gas identifier " + minor_gases({i}));
            if (available.exists(trimmed(gas_minor({idx}))))
            {
                keep.push_back(i);
                ncontrib += limits_gpt({2, i}) - limits_gpt({1, i}) + 1;
            }
        }
        const int nk = int(keep.size());
        const int neta = kminor.dim(2), ntemp = kminor.dim(3);
        Minor_tables t;
        t.gases.set_dims({nk}); t.scaling_gas.set_dims({nk}); t.limits_gpt.set_dims({2, nk});
        t.scales_with_density.set_dims({nk}); t.scale_by_complement.set_dims({nk}); t.kminor_start.set_dims({nk});
        t.kminor.set_dims({ntemp, neta, ncontrib});
        int next = 1;
        for (int k=1; k<=nk; ++k)
        {
            const int i = keep[k-1];
            const int ng = limits_gpt({2, i}) - limits_gpt({1, i}) + 1;
            t.gases({k}) = minor_gases({i}); t.scaling_gas({k}) = scaling_gas({i});
            t.limits_gpt({1, k}) = limits_gpt({1, i}); t.limits_gpt({2, k}) = limits_gpt({2, i});
            t.scales_with_density({k}) = scales_with_density({i}); t.scale_by_complement({k}) = scale_by_complement({i});
            t.kminor_start({k}) = next;
            for (int j=0; j<ng; ++j)
                for (int ie=1; ie<=neta; ++ie)
                    for (int it=1; it<=ntemp; ++it)
                        t.kminor({it, ie, next + j}) = kminor({kminor_start({i}) + j, ie, it});
            next += ng;
        }
        return t;
    }

    Array<int,1> minor_gas_index(const Array<std::string,1>& gas_names, const Array<std::string,1>& gas_minor,
                                 const Array<std::string,1>& identifier_minor, const Array<std::string,1>& minor_gases)
    {
        Array<int,1> idx({minor_gases.dim(1)});
        for (int i=1; i<=minor_gases.dim(1); ++i)
            idx({i}) = find_index(gas_names, trimmed(gas_minor({find_index(identifier_minor, minor_gases({i}))})));
        return idx;
    }

    Array<int,1> scaling_gas_index(const Array<std::string,1>& gas_names, const Array<std::string,1>& scaling_gas)
    {
        Array<int,1> idx({scaling_gas.dim(1)});
        for (int i=1; i<=scaling_gas.dim(1); ++i)
            idx({i}) = find_index(gas_names, trimmed(scaling_gas({i})));     // -1 (no scaling gas) is <= 0 for the kernels
        return idx;
    }
}





// Constructor of longwave variant.
Gas_optics_rrtmgp_gpu::Gas_optics_rrtmgp_gpu(
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
        const Array<Float,3>& rayl_upper) :
    Gas_optics_gpu(band_lims_wavenum, band2gpt),
    totplnk(totplnk)
{
    (void)temp_ref_p; (void)temp_ref_t;
    init_abs_coeffs(
            available_gases, gas_names, key_species, press_ref, temp_ref, press_ref_trop, vmr_ref,
            kmajor, kminor_lower, kminor_upper, gas_minor, identifier_minor, minor_gases_lower, minor_gases_upper,
            minor_limits_gpt_lower, minor_limits_gpt_upper, minor_scales_with_density_lower, minor_scales_with_density_upper,
            scaling_gas_lower, scaling_gas_upper, scale_by_complement_lower, scale_by_complement_upper,
            kminor_start_lower, kminor_start_upper, rayl_lower, rayl_upper);

    // Planck fraction: file order (ngpt, neta, npres+1, ntemp) -> kernel order (ntemp, neta, npres+1, ngpt)
    Array<Float,4> pf({planck_frac.dim(4), planck_frac.dim(2), planck_frac.dim(3), planck_frac.dim(1)});
    for (int i4=1; i4<=pf.dim(4); ++i4)
        for (int i3=1; i3<=pf.dim(3); ++i3)
            for (int i2=1; i2<=pf.dim(2); ++i2)
                for (int i1=1; i1<=pf.dim(1); ++i1)
                    pf({i1, i2, i3, i4}) = planck_frac({i4, i2, i3, i1});
    planck_frac_gpu = pf;
    totplnk_gpu = this->totplnk;

    // Temperature steps for Planck function interpolation: assumes equal spacing of the totplnk table
    this->totplnk_delta = (temp_ref_max - temp_ref_min) / (this->totplnk.dim(1) - 1);
}


// Constructor of the shortwave variant.
Gas_optics_rrtmgp_gpu::Gas_optics_rrtmgp_gpu(
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
        const Array<Float,3>& rayl_upper) :
    Gas_optics_gpu(band_lims_wavenum, band2gpt)
{
    (void)temp_ref_p; (void)temp_ref_t; (void)tsi_default;
    init_abs_coeffs(
            available_gases, gas_names, key_species, press_ref, temp_ref, press_ref_trop, vmr_ref,
            kmajor, kminor_lower, kminor_upper, gas_minor, identifier_minor, minor_gases_lower, minor_gases_upper,
            minor_limits_gpt_lower, minor_limits_gpt_upper, minor_scales_with_density_lower, minor_scales_with_density_upper,
            scaling_gas_lower, scaling_gas_upper, scale_by_complement_lower, scale_by_complement_upper,
            kminor_start_lower, kminor_start_upper, rayl_lower, rayl_upper);

    this->solar_source_quiet = solar_src_quiet;
    this->solar_source_facular = solar_src_facular;
    this->solar_source_sunspot = solar_src_sunspot;
    this->solar_source.set_dims(solar_src_quiet.get_dims());
    set_solar_variability(mg_default, sb_default);
}


void Gas_optics_rrtmgp_gpu::init_abs_coeffs(
        const Gas_concs_gpu& available_gases,
        const Array<std::string,1>& gas_names_in,
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
        const Array<Float,3>& rayl_upper)
{
    // ---- gases known to the k-distribution AND provided by the host model, in k-distribution order
    std::vector<std::string> use;
    for (const std::string& s : gas_names_in.v())
        if (available_gases.exists(s)) use.push_back(s);
    const int ngas = int(use.size());
    this->gas_names = Array<std::string,1>(use, {ngas});
    this->idx_h2o = find_index(this->gas_names, "h2o");
    if (this->idx_h2o < 0) throw std::runtime_error("idx_h2o cannot be found");

    // ---- reference mixing ratios with slot 0 = dry air: (2, 0:ngas, ntemp)
    Array<Float,3> vmr_red({vmr_ref.dim(1), ngas+1, vmr_ref.dim(3)});
    for (int i3=1; i3<=vmr_ref.dim(3); ++i3)
        for (int i1=1; i1<=vmr_ref.dim(1); ++i1)
        {
            vmr_red({i1, 1, i3}) = vmr_ref({i1, 1, i3});
            for (int ig=1; ig<=ngas; ++ig)
                vmr_red({i1, ig+1, i3}) = vmr_ref({i1, find_index(gas_names_in, this->gas_names({ig}))+1, i3});
        }
    vmr_ref_gpu = vmr_red;

    // ---- minor gases
    Minor_tables lo = reduce_minor(available_gases, gas_minor, identifier_minor, kminor_lower, minor_gases_lower,
            minor_limits_gpt_lower, minor_scales_with_density_lower, scaling_gas_lower, scale_by_complement_lower, kminor_start_lower);
    Minor_tables up = reduce_minor(available_gases, gas_minor, identifier_minor, kminor_upper, minor_gases_upper,
            minor_limits_gpt_upper, minor_scales_with_density_upper, scaling_gas_upper, scale_by_complement_upper, kminor_start_upper);
    nminorlower = lo.gases.dim(1); nminorklower = lo.kminor.dim(3);
    nminorupper = up.gases.dim(1); nminorkupper = up.kminor.dim(3);
    kminor_lower_gpu = lo.kminor; kminor_upper_gpu = up.kminor;
    minor_limits_gpt_lower_gpu = lo.limits_gpt; minor_limits_gpt_upper_gpu = up.limits_gpt;
    minor_scales_with_density_lower_gpu = lo.scales_with_density; minor_scales_with_density_upper_gpu = up.scales_with_density;
    scale_by_complement_lower_gpu = lo.scale_by_complement; scale_by_complement_upper_gpu = up.scale_by_complement;
    kminor_start_lower_gpu = lo.kminor_start; kminor_start_upper_gpu = up.kminor_start;
    idx_minor_lower_gpu = minor_gas_index(this->gas_names, gas_minor, identifier_minor, lo.gases);
    idx_minor_upper_gpu = minor_gas_index(this->gas_names, gas_minor, identifier_minor, up.gases);
    idx_minor_scaling_lower_gpu = scaling_gas_index(this->gas_names, lo.scaling_gas);
    idx_minor_scaling_upper_gpu = scaling_gas_index(this->gas_names, up.scaling_gas);

    // ---- major absorption: file order (ngpt, neta, npres+1, ntemp) -> (ntemp, neta, npres+1, ngpt)
    ntemp = kmajor.dim(4); neta = kmajor.dim(2); npres = kmajor.dim(3) - 1;
    Array<Float,4> km({ntemp, neta, npres+1, kmajor.dim(1)});
    for (int i4=1; i4<=km.dim(4); ++i4)
        for (int i3=1; i3<=km.dim(3); ++i3)
            for (int i2=1; i2<=km.dim(2); ++i2)
                for (int i1=1; i1<=km.dim(1); ++i1)
                    km({i1, i2, i3, i4}) = kmajor({i4, i2, i3, i1});
    kmajor_gpu = km;

    // ---- Rayleigh: (ngpt, neta, ntemp) x {lower, upper} -> (ntemp, neta, ngpt, 2)
    has_rayleigh = rayl_lower.size() > 0;
    if (has_rayleigh)
    {
        Array<Float,4> kr({rayl_lower.dim(3), rayl_lower.dim(2), rayl_lower.dim(1), 2});
        for (int i3=1; i3<=kr.dim(3); ++i3)
            for (int i2=1; i2<=kr.dim(2); ++i2)
                for (int i1=1; i1<=kr.dim(1); ++i1)
                {
                    kr({i1, i2, i3, 1}) = rayl_lower({i3, i2, i1});
                    kr({i1, i2, i3, 2}) = rayl_upper({i3, i2, i1});
                }
        krayl_gpu = kr;
    }

    // ---- reference grids
    Array<Float,1> press_ref_log(press_ref);
    for (Float& p : press_ref_log.v()) p = std::log(p);
    press_ref_log_gpu = press_ref_log;
    temp_ref_gpu = temp_ref;
    press_ref_trop_log = std::log(press_ref_trop);
    temp_ref_min = temp_ref({1}); temp_ref_max = temp_ref({temp_ref.dim(1)});
    press_ref_min = press_ref({press_ref.dim(1)}); press_ref_max = press_ref({1});
    press_ref_log_delta = (std::log(press_ref_min) - std::log(press_ref_max)) / (press_ref.dim(1) - 1);
    temp_ref_delta = (temp_ref_max - temp_ref_min) / (temp_ref.dim(1) - 1);

    // ---- key species -> indices into the reduced gas list; flavors in order of first appearance (band-major,
    //      lower then upper atmosphere); a (0,0) pair means "no key species" and is rewritten to (2,2)
    const int nbnd = key_species.dim(3);
    Array<int,3> ks_red({2, 2, nbnd});
    for (int ib=1; ib<=nbnd; ++ib)
        for (int ia=1; ia<=2; ++ia)
            for (int ip=1; ip<=2; ++ip)
            {
                const int ks = key_species({ip, ia, ib});
                int r = 0;
                if (ks != 0)
                {
                    r = find_index(this->gas_names, gas_names_in({ks}));
                    if (r < 0) throw std::runtime_error("Gas optics: required gas " + gas_names_in({ks}) + " is missing");
                }
                ks_red({ip, ia, ib}) = r;
            }
    auto pair_of = [&](const int ia, const int ib)
    {
        std::array<int,2> p = { ks_red({1, ia, ib}), ks_red({2, ia, ib}) };
        if (p[0] == 0 && p[1] == 0) p = {2, 2};
        return p;
    };
    std::vector<std::array<int,2>> flav;
    for (int ib=1; ib<=nbnd; ++ib)
        for (int ia=1; ia<=2; ++ia)
        {
            const auto p = pair_of(ia, ib);
            if (std::find(flav.begin(), flav.end(), p) == flav.end()) flav.push_back(p);
        }
    this->flavor.set_dims({2, int(flav.size())});
    for (int i=1; i<=int(flav.size()); ++i) { this->flavor({1, i}) = flav[i-1][0]; this->flavor({2, i}) = flav[i-1][1]; }
    const Array<int,1> gpt2band = this->get_gpoint_bands();
    this->gpoint_flavor.set_dims({2, gpt2band.dim(1)});
    for (int ig=1; ig<=gpt2band.dim(1); ++ig)
        for (int ia=1; ia<=2; ++ia)
        {
            const auto p = pair_of(ia, gpt2band({ig}));
            this->gpoint_flavor({ia, ig}) = int(std::find(flav.begin(), flav.end(), p) - flav.begin()) + 1;
        }
    flavor_gpu = this->flavor;
    gpoint_flavor_gpu = this->gpoint_flavor;
}


void Gas_optics_rrtmgp_gpu::set_solar_variability(const Float mg_index, const Float sb_index)
{
    // /root/reference/src_cuda/Gas_optics_rrtmgp.cu:1204-1217
    constexpr Float a_offset = Float(0.1495954);
    constexpr Float b_offset = Float(0.00066696);
    for (int igpt=1; igpt<=this->solar_source_quiet.dim(1); ++igpt)
        this->solar_source({igpt}) = this->solar_source_quiet({igpt})
                + (mg_index - a_offset) * this->solar_source_facular({igpt})
                + (sb_index - b_offset) * this->solar_source_sunspot({igpt});
    this->solar_source_gpu = this->solar_source;
}


Float Gas_optics_rrtmgp_gpu::get_tsi() const
{
    Float tsi = 0.;
    for (int igpt=1; igpt<=this->solar_source.dim(1); ++igpt)
        tsi += this->solar_source({igpt});
    return tsi;
}


void Gas_optics_rrtmgp_gpu::get_col_dry(Array_gpu<Float,2>& col_dry, const Array_gpu<Float,2>& vmr_h2o, const Array_gpu<Float,2>& plev)
{
    RRX_CALL(rrx_get_col_dry, col_dry.dim(1), col_dry.dim(2), vmr_h2o.ptr(), plev.ptr(), col_dry.ptr());
}


// col_gas(ncol,nlay,0:ngas): slot 0 = col_dry, slot i = vmr_i*col_dry (src_cuda/Gas_optics_rrtmgp.cu:392-422,1023-1028)
void Gas_optics_rrtmgp_gpu::fill_col_gas(
        const int ncol, const int nlay, const Gas_concs_gpu& gas_desc, const Array_gpu<Float,2>& col_dry, Array_gpu<Float,3>& col_gas)
{
    const int ngas = this->gas_names.dim(1);
    col_gas.set_dims({ncol, nlay, ngas+1});
    if (ngas <= 32)                             // one launch for all gases
    {
        const Float* src[32]; int d1[32], d2[32];
        for (int igas=1; igas<=ngas; ++igas)
        {
            const Array_gpu<Float,2>& v = gas_desc.get_vmr(this->gas_names({igas}));
            src[igas-1] = v.ptr(); d1[igas-1] = v.dim(1); d2[igas-1] = v.dim(2);
        }
        RRX_CALL(rrx_fill_gases_all, ncol, nlay, ngas, src, d1, d2, col_gas.ptr(), col_dry.ptr());
        return;
    }
    Array_gpu<Float,3> vmr({ncol, nlay, ngas});
    for (int igas=0; igas<=ngas; ++igas)
    {
        const Array_gpu<Float,2>& vmr_2d = gas_desc.get_vmr(this->gas_names({igas > 0 ? igas : 1}));
        RRX_CALL(rrx_fill_gases, ncol, nlay, vmr_2d.dim(1), vmr_2d.dim(2), ngas, igas, vmr.ptr(), vmr_2d.ptr(), col_gas.ptr(), col_dry.ptr());
    }
}

// The "direct" entry points compute the interpolation state (interpolation_kernel of the reference) inside its consumers:
// same expressions and bits as interpolation -> compute_tau_absorption -> ..., without the seven intermediate arrays.
#define RRX_MINOR_ARGS \
        ncol, nlay, this->get_nband(), this->get_ngpt(), ngas, this->get_nflav(), neta, npres, ntemp, \
        nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o, \
        gpoint_flavor_gpu.ptr(), this->get_band_lims_gpoint_gpu().ptr(), kmajor_gpu.ptr(), kminor_lower_gpu.ptr(), kminor_upper_gpu.ptr(), \
        minor_limits_gpt_lower_gpu.ptr(), minor_limits_gpt_upper_gpu.ptr(), \
        minor_scales_with_density_lower_gpu.ptr(), minor_scales_with_density_upper_gpu.ptr(), \
        scale_by_complement_lower_gpu.ptr(), scale_by_complement_upper_gpu.ptr(), \
        idx_minor_lower_gpu.ptr(), idx_minor_upper_gpu.ptr(), idx_minor_scaling_lower_gpu.ptr(), idx_minor_scaling_upper_gpu.ptr(), \
        kminor_start_lower_gpu.ptr(), kminor_start_upper_gpu.ptr()
#define RRX_INTERP_ARGS \
        flavor_gpu.ptr(), press_ref_log_gpu.ptr(), temp_ref_gpu.ptr(), press_ref_log_delta, temp_ref_min, temp_ref_delta, press_ref_trop_log, \
        vmr_ref_gpu.ptr()


// Gas optics solver longwave variant.
void Gas_optics_rrtmgp_gpu::gas_optics(
        const Array_gpu<Float,2>& play,
        const Array_gpu<Float,2>& plev,
        const Array_gpu<Float,2>& tlay,
        const Array_gpu<Float,1>& tsfc,
        const Gas_concs_gpu& gas_desc,
        std::unique_ptr<Optical_props_arry_gpu>& optical_props,
        Source_func_lw_gpu& sources,
        const Array_gpu<Float,2>& col_dry,
        const Array_gpu<Float,2>& tlev,
        const Optical_props_1scl_gpu* add_by_band)
{
    (void)plev;
    if (add_by_band != nullptr && add_by_band->get_ngpt() != this->get_nband())
        throw std::runtime_error("Cannot add optical properties with incompatible band - gpoint combination");
    const Float* by_band_tau = add_by_band ? add_by_band->get_tau().ptr() : nullptr;
    const int ncol = play.dim(1);
    const int nlay = play.dim(2);
    const int ngas = this->gas_names.dim(1);
    Array_gpu<Float,3> col_gas;
    fill_col_gas(ncol, nlay, gas_desc, col_dry, col_gas);

    // the one synchronous 1-element read-back of the reference (src_cuda/Gas_optics_rrtmgp.cu:1190), unless the caller has
    // stated the vertical ordering (set_vertical_ordering): then the whole call is asynchronous on the current stream
    const int sfc_lay = (vertical_ordering < 0) ? (play({1, 1}) > play({1, nlay}) ? 1 : nlay) : (vertical_ordering == 1 ? nlay : 1);
    if (sources.planck_lite_wanted())
    {
        // optical depths + Planck fractions + band Planck functions in one pass; the broadband solver forms the sources
        RRX_CALL(rrx_gas_optics_lw_fractions_allsky,
                ncol, nlay, this->get_nband(), this->get_ngpt(), ngas, this->get_nflav(), neta, npres, ntemp, this->get_nPlanckTemp(),
                nminorlower, nminorklower, nminorupper, nminorkupper, idx_h2o,
                gpoint_flavor_gpu.ptr(), this->get_band_lims_gpoint_gpu().ptr(), this->get_gpoint_bands_gpu().ptr(),
                kmajor_gpu.ptr(), kminor_lower_gpu.ptr(), kminor_upper_gpu.ptr(),
                minor_limits_gpt_lower_gpu.ptr(), minor_limits_gpt_upper_gpu.ptr(),
                minor_scales_with_density_lower_gpu.ptr(), minor_scales_with_density_upper_gpu.ptr(),
                scale_by_complement_lower_gpu.ptr(), scale_by_complement_upper_gpu.ptr(),
                idx_minor_lower_gpu.ptr(), idx_minor_upper_gpu.ptr(), idx_minor_scaling_lower_gpu.ptr(), idx_minor_scaling_upper_gpu.ptr(),
                kminor_start_lower_gpu.ptr(), kminor_start_upper_gpu.ptr(), RRX_INTERP_ARGS,
                play.ptr(), tlay.ptr(), tlev.ptr(), tsfc.ptr(), sfc_lay, col_gas.ptr(),
                planck_frac_gpu.ptr(), totplnk_delta, totplnk_gpu.ptr(),
                optical_props->get_tau().ptr(), sources.get_planck_frac().ptr(), sources.get_planck_lay().ptr(), sources.get_planck_lev().ptr(),
                sources.get_sfc_source().ptr(), sources.get_sfc_source_jac().ptr(), by_band_tau);
        sources.set_fractions_valid(true);
        return;
    }
    RRX_CALL(rrx_gas_optics_lw_direct_allsky, RRX_MINOR_ARGS, RRX_INTERP_ARGS, play.ptr(), tlay.ptr(), col_gas.ptr(), optical_props->get_tau().ptr(), by_band_tau);
    sources.ensure_full_arrays();
    sources.set_fractions_valid(false);
    RRX_CALL(rrx_planck_source_direct,
            ncol, nlay, this->get_nband(), this->get_ngpt(), ngas, this->get_nflav(), neta, npres, ntemp, this->get_nPlanckTemp(),
            play.ptr(), tlay.ptr(), tlev.ptr(), tsfc.ptr(), sfc_lay, col_gas.ptr(), RRX_INTERP_ARGS,
            this->get_gpoint_bands_gpu().ptr(), this->get_band_lims_gpoint_gpu().ptr(), planck_frac_gpu.ptr(),
            totplnk_delta, totplnk_gpu.ptr(), gpoint_flavor_gpu.ptr(),
            sources.get_sfc_source().ptr(), sources.get_lay_source().ptr(), sources.get_lev_source().ptr(), sources.get_sfc_source_jac().ptr());
}


// Gas optics solver shortwave variant.
void Gas_optics_rrtmgp_gpu::gas_optics(
        const Array_gpu<Float,2>& play,
        const Array_gpu<Float,2>& plev,
        const Array_gpu<Float,2>& tlay,
        const Gas_concs_gpu& gas_desc,
        std::unique_ptr<Optical_props_arry_gpu>& optical_props,
        Array_gpu<Float,2>& toa_src,
        const Array_gpu<Float,2>& col_dry,
        const Optical_props_2str_gpu* add_by_band)
{
    (void)plev;
    if (add_by_band != nullptr && add_by_band->get_ngpt() != this->get_nband())
        throw std::runtime_error("Cannot add optical properties with incompatible band - gpoint combination");
    const int ncol = play.dim(1);
    const int nlay = play.dim(2);
    const int ngas = this->gas_names.dim(1);
    Array_gpu<Float,3> col_gas;
    fill_col_gas(ncol, nlay, gas_desc, col_dry, col_gas);
    // absorption + Rayleigh + combine in one pass over the output (same arithmetic as the three reference launchers)
    if (add_by_band != nullptr)
    {
        // all-sky: gas + by-band properties combined where they are stored; g is a real array from here on
        Optical_props_2str_gpu& op = dynamic_cast<Optical_props_2str_gpu&>(*optical_props);
        op.forget_g_zero();
        RRX_CALL(rrx_gas_optics_sw_direct_allsky, RRX_MINOR_ARGS, RRX_INTERP_ARGS, play.ptr(), tlay.ptr(), col_gas.ptr(), col_dry.ptr(), krayl_gpu.ptr(),
                op.get_tau().ptr(), op.get_ssa().ptr(), op.get_g().ptr(),
                add_by_band->get_tau().ptr(), add_by_band->get_ssa().ptr(), add_by_band->get_g().ptr());
    }
    else
    {
        RRX_CALL(rrx_gas_optics_sw_direct, RRX_MINOR_ARGS, RRX_INTERP_ARGS, play.ptr(), tlay.ptr(), col_gas.ptr(), col_dry.ptr(), krayl_gpu.ptr(),
                optical_props->get_tau().ptr(), optical_props->get_ssa().ptr(), static_cast<Float*>(nullptr));
        optical_props->set_g_zero();          // g == 0: not written; materialised on the first get_g() (clouds, output)
    }
    // External source function is constant in the column.
    RRX_CALL(rrx_spread_col, ncol, this->get_ngpt(), toa_src.ptr(), solar_source_gpu.ptr());
}
