// C entry points around Radiation_solver_longwave / _shortwave for a foreign host (bench.py --driver cxx, tests): the reference's
// class structure -- Gas_concs_gpu, Gas_optics_rrtmgp_gpu, Cloud_optics_gpu, Rte_lw_gpu / Rte_sw_gpu, Fluxes_broadband_gpu inside
// solve_gpu (include_test/Radiation_solver.h = /root/reference/include_test/Radiation_solver.h:33-235) -- driven on device arrays the
// caller owns. Coefficients come from the files the reference's driver reads (coefficients_lw.nc, ...); the atmosphere is handed
// over as device pointers in the Array layout (column index fastest). Nothing here synchronises: rrx_cxx_driver_solve enqueues one
// LW + SW solve on the given stream and returns.
#include <memory>
#include <string>
#include "Radiation_solver.h"
#include "rrx_cxx_driver.h"

namespace
{
    thread_local std::string g_error;

    struct Driver
    {
        std::unique_ptr<Radiation_solver_longwave> lw;
        std::unique_ptr<Radiation_solver_shortwave> sw;
        Gas_concs_gpu gases;
        bool clouds = false;
        int ncol = 0, nlay = 0;
        // broadband outputs of the last solve, (ncol, nlay+1) each: LW up, dn, net; SW up, dn, dn_dir, net
        Array_gpu<Float,2> lw_up, lw_dn, lw_net, sw_up, sw_dn, sw_dir, sw_net;
    };

    template<typename Fn> int guarded(Fn&& f)
    {
        try { f(); return 0; }
        catch (const std::exception& e) { g_error = e.what(); return 1; }
    }
    Array_gpu<Float,2> view2(const Float* p, const int n1, const int n2) { return p ? Array_gpu<Float,2>(const_cast<Float*>(p), {n1, n2}) : Array_gpu<Float,2>(); }
    Array_gpu<Float,1> view1(const Float* p, const int n1) { return p ? Array_gpu<Float,1>(const_cast<Float*>(p), {n1}) : Array_gpu<Float,1>(); }
}

extern "C"
{
const char* rrx_cxx_driver_error() { return g_error.c_str(); }

// gas_names: the gases the caller will provide (rrx_cxx_driver_set_gas), needed when the k-distributions are loaded
void* rrx_cxx_driver_create(const char* dir, const int ngas, const char* const* gas_names, const int clouds, const int top_at_1)
{
    Driver* d = nullptr;
    const int rc = guarded([&]
    {
        const std::string base = std::string(dir) + "/";
        std::unique_ptr<Driver> drv = std::make_unique<Driver>();
        Gas_concs host_gases;                                       // (names only: availability is what the constructors look at)
        for (int i=0; i<ngas; ++i) host_gases.set_vmr(gas_names[i], Float(0.));
        drv->gases = Gas_concs_gpu(host_gases);
        drv->clouds = clouds != 0;
        drv->lw = std::make_unique<Radiation_solver_longwave>(drv->gases, base + "coefficients_lw.nc", clouds ? base + "cloud_coefficients_lw.nc" : std::string());
        drv->sw = std::make_unique<Radiation_solver_shortwave>(drv->gases, clouds != 0, false, base + "coefficients_sw.nc",
                                                               clouds ? base + "cloud_coefficients_sw.nc" : std::string(), std::string());
        drv->lw->set_vertical_ordering(top_at_1); drv->sw->set_vertical_ordering(top_at_1);       // stated: no read-backs inside solve_gpu
        d = drv.release();
    });
    return rc == 0 ? d : nullptr;
}

void rrx_cxx_driver_destroy(void* h) { delete static_cast<Driver*>(h); }

// vmr: device pointer to a (n1, n2) array in the Array layout: (1,1) scalar, (1,nlay) profile or (ncol,nlay) field; the values are COPIED (call again when they change)
int rrx_cxx_driver_set_gas(void* h, const char* name, const Float* vmr, const int n1, const int n2)
{
    return guarded([&] { static_cast<Driver*>(h)->gases.set_vmr(name, view2(vmr, n1, n2)); });
}

int rrx_cxx_driver_settings(void* h, const int column_block, const int broadband, const int sort_mode, const int pad)
{
    return guarded([&]
    {
        Driver& d = *static_cast<Driver*>(h);
        d.lw->set_column_block(column_block); d.sw->set_column_block(column_block);
        d.lw->set_broadband_solvers(broadband != 0); d.sw->set_broadband_solvers(broadband != 0);
        d.lw->set_column_sorting(sort_mode); d.sw->set_column_sorting(sort_mode);
        d.lw->set_column_padding(pad != 0); d.sw->set_column_padding(pad != 0);
    });
}

// One LW + SW solve (fluxes only) on `stream`. Arrays: (ncol,nlay) / (ncol,nlay+1) fields, (ncol) vectors, surface properties (nbnd,ncol);
// lwp, iwp, rel, dei may be NULL without clouds. out7: seven device arrays (ncol, nlay+1) for LW up, dn, net and SW up, dn, dn_dir, net,
// or NULL: the driver keeps them (rrx_cxx_driver_fluxes).
int rrx_cxx_driver_solve(void* h, const int ncol, const int nlay, const int nbnd_lw, const int nbnd_sw,
        const Float* p_lay, const Float* p_lev, const Float* t_lay, const Float* t_lev, const Float* t_sfc,
        const Float* emis_sfc, const Float* sfc_alb_dir, const Float* sfc_alb_dif, const Float* tsi_scaling, const Float* mu0,
        const Float* lwp, const Float* iwp, const Float* rel, const Float* dei, Float* const* out7, void* stream)
{
    return guarded([&]
    {
        Driver& d = *static_cast<Driver*>(h);
        rrx_host::set_stream(stream);
        const int nlev = nlay + 1;
        Array_gpu<Float,2>* outs[7] = {&d.lw_up, &d.lw_dn, &d.lw_net, &d.sw_up, &d.sw_dn, &d.sw_dir, &d.sw_net};
        if (out7 != nullptr)                      // the caller's seven (ncol, nlay+1) arrays: the solvers write there
        {
            for (int i=0; i<7; ++i) *outs[i] = Array_gpu<Float,2>(out7[i], {ncol, nlev});
            d.ncol = ncol; d.nlay = nlay;
        }
        else if (d.ncol != ncol || d.nlay != nlay)
        {
            for (Array_gpu<Float,2>* a : outs) { *a = Array_gpu<Float,2>(); a->set_dims({ncol, nlev}); }
            d.ncol = ncol; d.nlay = nlay;
        }
        const Array_gpu<Float,2> pl = view2(p_lay, ncol, nlay), pv = view2(p_lev, ncol, nlev), tl = view2(t_lay, ncol, nlay), tv = view2(t_lev, ncol, nlev);
        const Array_gpu<Float,2> c_lwp = view2(lwp, ncol, nlay), c_iwp = view2(iwp, ncol, nlay), c_rel = view2(rel, ncol, nlay), c_dei = view2(dei, ncol, nlay);
        const Array_gpu<Float,2> no_col_dry, no_rh;
        Array_gpu<Float,3> o3a, o3b, o3c, b1, b2, b3, b4; Array_gpu<Float,2> o2;
        d.lw->solve_gpu(true, d.clouds, false, false, d.gases, pl, pv, tl, tv, no_col_dry, view1(t_sfc, ncol), view2(emis_sfc, nbnd_lw, ncol),
                        c_lwp, c_iwp, c_rel, c_dei, o3a, o3b, o3c, o2, d.lw_up, d.lw_dn, d.lw_net, b1, b2, b3);
        Aerosol_concs_gpu no_aerosols;
        d.sw->solve_gpu(true, d.clouds, false, false, false, true, false, d.gases, pl, pv, tl, tv, no_col_dry,
                        view2(sfc_alb_dir, nbnd_sw, ncol), view2(sfc_alb_dif, nbnd_sw, ncol), view1(tsi_scaling, ncol), view1(mu0, ncol),
                        c_lwp, c_iwp, c_rel, c_dei, no_rh, no_aerosols, o3a, o3b, o3c, o2, d.sw_up, d.sw_dn, d.sw_dir, d.sw_net, b1, b2, b3, b4);
    });
}

// device pointers of the seven broadband flux arrays of the last solve, (ncol, nlay+1) each
int rrx_cxx_driver_fluxes(void* h, const Float** ptrs)
{
    return guarded([&]
    {
        Driver& d = *static_cast<Driver*>(h);
        const Array_gpu<Float,2>* a[7] = {&d.lw_up, &d.lw_dn, &d.lw_net, &d.sw_up, &d.sw_dn, &d.sw_dir, &d.sw_net};
        for (int i=0; i<7; ++i) ptrs[i] = a[i]->ptr();
    });
}
}
