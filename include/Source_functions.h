/* Source_func_lw_gpu -- interface of /root/reference/include/Source_functions.h:66-93.
 * Addition (same idea as the lazy g == 0 of Optical_props_2str_gpu): in "Planck-lite" mode (enable_planck_lite(true), set by
 * Radiation_solver_longwave when the solver runs in broadband mode) Gas_optics_rrtmgp_gpu::gas_optics stores the Planck
 * fractions pfrac(ncol,nlay,ngpt) and the band Planck functions B_lay(ncol,nlay,nbnd), B_lev(ncol,nlay+1,nbnd) instead of the two
 * products; Rte_lw_gpu::rte_lw hands them to rrx_lw_solver_noscat_fractions, which forms lay_source and lev_source on the fly.
 * Any access through get_lay_source() / get_lev_source() materialises the arrays first (rrx_planck_sources_from_fractions,
 * bit-identical to Planck_source_kernel), so the reference semantics are unchanged for every other caller. */
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
        Array_gpu<Float,3>& get_lay_source()     { materialize(); return lay_source; }
        Array_gpu<Float,3>& get_lev_source()     { materialize(); return lev_source; }
        const Array_gpu<Float,2>& get_sfc_source()     const { return sfc_source;     }
        const Array_gpu<Float,2>& get_sfc_source_jac() const { return sfc_source_jac; }
        const Array_gpu<Float,3>& get_lay_source()     const { materialize(); return lay_source; }
        const Array_gpu<Float,3>& get_lev_source()     const { materialize(); return lev_source; }

        // Planck-lite mode
        void enable_planck_lite(const bool on) { lite_wanted = on; }
        bool planck_lite_wanted() const { return lite_wanted; }
        bool holds_fractions() const { return fractions_valid; }           // true between a lite gas_optics() and the next materialisation
        void set_fractions_valid(const bool v) { fractions_valid = v; }
        Array_gpu<Float,3>& get_planck_frac();                              // allocated on first use
        Array_gpu<Float,3>& get_planck_lay();
        Array_gpu<Float,3>& get_planck_lev();
        const Array_gpu<Float,3>& get_planck_frac() const { return pfrac; }
        const Array_gpu<Float,3>& get_planck_lay() const { return blay; }
        const Array_gpu<Float,3>& get_planck_lev() const { return blev; }
        void ensure_full_arrays();                                          // lay_source / lev_source allocated (for the full Planck kernel)
    private:
        void materialize() const;
        int n_col, n_lay;
        Array_gpu<Float,2> sfc_source;
        Array_gpu<Float,2> sfc_source_jac;
        mutable Array_gpu<Float,3> lay_source;       // allocated on first use
        mutable Array_gpu<Float,3> lev_source;
        Array_gpu<Float,3> pfrac, blay, blev;
        bool lite_wanted = false;
        mutable bool fractions_valid = false;
};
#endif
