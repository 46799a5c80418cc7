/* Optical_props_gpu and its 1scl / 2str array forms -- interface of /root/reference/include/Optical_props.h:175-294.
 * Band structure: band2gpt(2,nband) 1-based inclusive g-point limits, gpt2band(ngpt) 1-based band of each g-point
 * (/root/reference/src/Optical_props.cpp:31-72). */
#ifndef OPTICAL_PROPS_H
#define OPTICAL_PROPS_H
#include <memory>
#include <stdexcept>
#include "Array.h"

class Optical_props_1scl_gpu;
class Optical_props_2str_gpu;
void add_to(Optical_props_1scl_gpu& op_inout, const Optical_props_1scl_gpu& op_in);
void add_to(Optical_props_2str_gpu& op_inout, const Optical_props_2str_gpu& op_in);

class Optical_props_gpu
{
    public:
        Optical_props_gpu(const Array<Float,2>& band_lims_wvn, const Array<int,2>& band_lims_gpt);
        Optical_props_gpu(const Array<Float,2>& band_lims_wvn);
        virtual ~Optical_props_gpu() {}
        Optical_props_gpu(const Optical_props_gpu&) = default;

        Array<int,1> get_gpoint_bands() const { return this->gpt2band; }
        const Array_gpu<int,1>& get_gpoint_bands_gpu() const { return this->gpt2band_gpu; }
        const Array_gpu<int,2>& get_band_lims_gpoint_gpu() const { return this->band2gpt_gpu; }
        int get_nband() const { return this->band2gpt.dim(2); }
        int get_ngpt() const { return this->band2gpt.max(); }
        const Array<int,2>& get_band_lims_gpoint() const { return this->band2gpt; }
        const Array<Float,2>& get_band_lims_wavenumber() const { return this->band_lims_wvn; }

    private:
        Array<int,2> band2gpt;
        Array_gpu<int,2> band2gpt_gpu;
        Array<int,1> gpt2band;
        Array_gpu<int,1> gpt2band_gpu;
        Array<Float,2> band_lims_wvn;
};

class Optical_props_arry_gpu : public Optical_props_gpu
{
    public:
        Optical_props_arry_gpu(const Optical_props_gpu& optical_props_gpu) : Optical_props_gpu(optical_props_gpu) {}
        virtual ~Optical_props_arry_gpu() {}
        virtual Array_gpu<Float,3>& get_tau() = 0;
        virtual Array_gpu<Float,3>& get_ssa() = 0;
        virtual Array_gpu<Float,3>& get_g  () = 0;
        virtual const Array_gpu<Float,3>& get_tau() const = 0;
        virtual const Array_gpu<Float,3>& get_ssa() const = 0;
        virtual const Array_gpu<Float,3>& get_g  () const = 0;
        virtual void delta_scale(const Array_gpu<Float,3>& forward_frac=Array_gpu<Float,3>()) = 0;
        virtual int get_ncol() const = 0;
        virtual int get_nlay() const = 0;
        // Addition: clear-sky gas optics produce g == 0 everywhere. set_g_zero() records that instead of writing the
        // array; get_g_or_null() hands nullptr to kernels that take "no asymmetry" natively (rrx_sw_solver_2stream);
        // any other access through get_g() fills the array with zeros first, so the reference semantics are unchanged.
        virtual void set_g_zero() {}
        virtual const Float* get_g_or_null() const { return get_g().ptr(); }
};

class Optical_props_1scl_gpu : public Optical_props_arry_gpu
{
    public:
        Optical_props_1scl_gpu(const int ncol, const int nlay, const Optical_props_gpu& optical_props_gpu);
        int get_ncol() const { return tau.dim(1); }
        int get_nlay() const { return tau.dim(2); }
        Array_gpu<Float,3>& get_tau() { return tau; }
        Array_gpu<Float,3>& get_ssa() { throw std::runtime_error("ssa is not available in this class"); }
        Array_gpu<Float,3>& get_g  () { throw std::runtime_error("g is available in this class"); }
        const Array_gpu<Float,3>& get_tau() const { return tau; }
        const Array_gpu<Float,3>& get_ssa() const { throw std::runtime_error("ssa is not available in this class"); }
        const Array_gpu<Float,3>& get_g  () const { throw std::runtime_error("g is available in this class"); }
        void delta_scale(const Array_gpu<Float,3>& forward_frac=Array_gpu<Float,3>()) {}
    private:
        Array_gpu<Float,3> tau;
};

class Optical_props_2str_gpu : public Optical_props_arry_gpu
{
    public:
        Optical_props_2str_gpu(const int ncol, const int nlay, const Optical_props_gpu& optical_props_gpu);
        int get_ncol() const { return tau.dim(1); }
        int get_nlay() const { return tau.dim(2); }
        Array_gpu<Float,3>& get_tau() { return tau; }
        Array_gpu<Float,3>& get_ssa() { return ssa; }
        Array_gpu<Float,3>& get_g  () { materialize_g(); return g; }
        const Array_gpu<Float,3>& get_tau() const { return tau; }
        const Array_gpu<Float,3>& get_ssa() const { return ssa; }
        const Array_gpu<Float,3>& get_g  () const { materialize_g(); return g; }
        void delta_scale(const Array_gpu<Float,3>& forward_frac=Array_gpu<Float,3>());
        void set_g_zero() { g_zero = true; }
        void forget_g_zero() { g_zero = false; }      // the array behind g is about to be written in full
        const Float* get_g_or_null() const { return g_zero ? nullptr : g.ptr(); }
    private:
        void materialize_g() const;
        Array_gpu<Float,3> tau;
        Array_gpu<Float,3> ssa;
        mutable Array_gpu<Float,3> g;
        mutable bool g_zero = false;
};
#endif
