/*
 * Radiation_solver_longwave / Radiation_solver_shortwave -- GPU solve path with the constructor and solve_gpu
 * argument lists of /root/reference/include_test/Radiation_solver.h:33-235, so that test_rte_rrtmgp-style drivers and a
 * host model (MicroHH) call it unchanged. Implementation: rte-rrtmgp-cpp_amd/host/src_test/Radiation_solver.cpp.
 *
 * Differences from the reference, all on the performance side:
 *   - the column block is a run-time setting (set_column_block; default 16384 instead of a fixed 1024): MI355X has
 *     288 GB of HBM, and bigger blocks mean fewer, larger launches;
 *   - block-sized workspaces (optical props, sources, g-point fluxes) are cached across calls;
 *   - without --output-bnd-fluxes the solvers can run in broadband mode (set_broadband_solvers(true)), the CPU path's
 *     convention (src_test/Radiation_solver.cpp:518-527), which never materialises per-g-point fluxes.
 */
#ifndef RADIATION_SOLVER_H
#define RADIATION_SOLVER_H
#include <memory>
#include <string>
#include "Array.h"
#include "Gas_concs.h"
#include "Gas_optics_rrtmgp.h"
#include "Cloud_optics.h"
#include "Aerosol_optics.h"
#include "Optical_props.h"
#include "Source_functions.h"
#include "Fluxes.h"
#include "Rte_lw.h"
#include "Rte_sw.h"

// Layer heating rates [K/s] from net (down - up) broadband fluxes and level pressures, on the device:
// -(g/cp) dF_net/dp with g = 9.80665 m s-2, cp = 1004.64 J kg-1 K-1 (dry air). flux_net, p_lev: (ncol, nlay+1); out: (ncol, nlay).
void compute_heating_rate(const Array_gpu<Float,2>& flux_net, const Array_gpu<Float,2>& p_lev, Array_gpu<Float,2>& heating_rate);

class Radiation_solver_longwave
{
    public:
        Radiation_solver_longwave(
                const Gas_concs_gpu& gas_concs,
                const std::string& file_name_gas,
                const std::string& file_name_cloud);

        void solve_gpu(
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
                Array_gpu<Float,3>& lw_bnd_flux_up, Array_gpu<Float,3>& lw_bnd_flux_dn, Array_gpu<Float,3>& lw_bnd_flux_net);

        int get_n_gpt_gpu() const { return this->kdist_gpu->get_ngpt(); }
        int get_n_bnd_gpu() const { return this->kdist_gpu->get_nband(); }
        Array<int,2> get_band_lims_gpoint_gpu() const { return this->kdist_gpu->get_band_lims_gpoint(); }
        Array<Float,2> get_band_lims_wavenumber_gpu() const { return this->kdist_gpu->get_band_lims_wavenumber(); }

        void set_column_block(const int n) { n_col_block = n; }
        void set_broadband_solvers(const bool b) { broadband_solvers = b; }
        // Host-model coupling: the solver object is persistent -- k-distribution and LUTs stay on the device, block
        // workspaces are cached across calls -- and with the vertical ordering stated (0 = surface first, 1 = top first; -1 =
        // detect with the reference's synchronous read-backs) solve_gpu() enqueues everything on the calling thread's stream
        // (rrx_host::set_stream) without synchronising, so a host model can overlap it with its own work.
        void set_vertical_ordering(const int top_at_1) { vertical_ordering = top_at_1; kdist_gpu->set_vertical_ordering(top_at_1); }
        // Column order of a solve (round 4; DESIGN "columns that differ"). Columns are independent, so solve_gpu may process them in
        // another order and on a padded count: sorted by surface pressure (neighbouring columns then share LUT boxes in the windowed
        // gas optics: 21 -> 13.5 ms per LW+SW solve at +-35 % pressure spread) and padded to a multiple of 16 columns (rows of the cell
        // arrays on 128-B lines: 16 385 columns cost 20 % more than 16 384 otherwise). Inputs are gathered on the device at the top
        // of the solve, outputs scattered back: the caller sees its own order. mode: 1 = always sort, 0 = never, -1 (default) = sort
        // when the surface pressure varies by more than 20 % inside some run of 256 columns -- decided ONCE per solver object, at its
        // first solve, with one synchronous read-back of a flag (a host model that must never synchronise states 0 or 1).
        // Not applied when optical properties are output (switch_output_optical).
        void set_column_sorting(const int mode) { column_sorting = mode; sort_decided = -1; }
        void set_column_padding(const bool b) { column_padding = b; }

    private:
        int column_sorting = -1, sort_decided = -1;
        bool column_padding = true, reordered_call = false;
        std::unique_ptr<Gas_optics_rrtmgp_gpu> kdist_gpu;
        std::unique_ptr<Cloud_optics_gpu> cloud_optics_gpu;
        Rte_lw_gpu rte_lw;
        int vertical_ordering = -1;
        int n_col_block = 16384;
        bool broadband_solvers = true;

        struct Workspace;
        std::shared_ptr<Workspace> ws_block, ws_residual;
};

class Radiation_solver_shortwave
{
    public:
        Radiation_solver_shortwave(
                const Gas_concs_gpu& gas_concs,
                const bool switch_cloud_optics,
                const bool switch_aerosol_optics,
                const std::string& file_name_gas,
                const std::string& file_name_cloud,
                const std::string& file_name_aerosol);

        void solve_gpu(
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
                Array_gpu<Float,3>& sw_bnd_flux_dn_dir, Array_gpu<Float,3>& sw_bnd_flux_net);

        int get_n_gpt_gpu() const { return this->kdist_gpu->get_ngpt(); }
        int get_n_bnd_gpu() const { return this->kdist_gpu->get_nband(); }
        Float get_tsi_gpu() const { return this->kdist_gpu->get_tsi(); }
        Array<int,2> get_band_lims_gpoint_gpu() const { return this->kdist_gpu->get_band_lims_gpoint(); }
        Array<Float,2> get_band_lims_wavenumber_gpu() const { return this->kdist_gpu->get_band_lims_wavenumber(); }

        void set_column_block(const int n) { n_col_block = n; }
        void set_broadband_solvers(const bool b) { broadband_solvers = b; }
        void set_vertical_ordering(const int top_at_1) { vertical_ordering = top_at_1; kdist_gpu->set_vertical_ordering(top_at_1); }
        // column order of a solve: see Radiation_solver_longwave
        void set_column_sorting(const int mode) { column_sorting = mode; sort_decided = -1; }
        void set_column_padding(const bool b) { column_padding = b; }

    private:
        int column_sorting = -1, sort_decided = -1;
        bool column_padding = true, reordered_call = false;
        std::unique_ptr<Gas_optics_rrtmgp_gpu> kdist_gpu;
        std::unique_ptr<Cloud_optics_gpu> cloud_optics_gpu;
        std::unique_ptr<Aerosol_optics_gpu> aerosol_optics_gpu;
        Rte_sw_gpu rte_sw;
        int vertical_ordering = -1;
        int n_col_block = 16384;
        bool broadband_solvers = true;

        struct Workspace;
        std::shared_ptr<Workspace> ws_block, ws_residual;
};
#endif
