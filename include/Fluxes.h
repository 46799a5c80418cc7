/* Fluxes_broadband_gpu / Fluxes_byband_gpu -- interface of /root/reference/include/Fluxes.h:127-210.
 * By-band reduction follows the Fortran kernels (spectral input, inclusive limits), not the buggy CUDA text. */
#ifndef FLUXES_H
#define FLUXES_H
#include <memory>
#include <stdexcept>
#include "Array.h"
#include "Optical_props.h"

class Fluxes_gpu
{
    public:
        virtual ~Fluxes_gpu() {}
        virtual void reduce(const Array_gpu<Float,3>& gpt_flux_up, const Array_gpu<Float,3>& gpt_flux_dn,
                const std::unique_ptr<Optical_props_arry_gpu>& optical_props, const Bool top_at_1) = 0;
        virtual void reduce(const Array_gpu<Float,3>& gpt_flux_up, const Array_gpu<Float,3>& gpt_flux_dn,
                const Array_gpu<Float,3>& gpt_flux_dn_dir,
                const std::unique_ptr<Optical_props_arry_gpu>& optical_props, const Bool top_at_1) = 0;
};

class Fluxes_broadband_gpu : public Fluxes_gpu
{
    public:
        Fluxes_broadband_gpu(const int ncol, const int nlev);
        virtual ~Fluxes_broadband_gpu() {}
        virtual void reduce(const Array_gpu<Float,3>& gpt_flux_up, const Array_gpu<Float,3>& gpt_flux_dn,
                const std::unique_ptr<Optical_props_arry_gpu>& optical_props, const Bool top_at_1);
        virtual void reduce(const Array_gpu<Float,3>& gpt_flux_up, const Array_gpu<Float,3>& gpt_flux_dn,
                const Array_gpu<Float,3>& gpt_flux_dn_dir,
                const std::unique_ptr<Optical_props_arry_gpu>& optical_props, const Bool top_at_1);
        Array_gpu<Float,2>& get_flux_up    () { return flux_up;     }
        Array_gpu<Float,2>& get_flux_dn    () { return flux_dn;     }
        Array_gpu<Float,2>& get_flux_dn_dir() { return flux_dn_dir; }
        Array_gpu<Float,2>& get_flux_net   () { return flux_net;    }
        virtual Array_gpu<Float,3>& get_bnd_flux_up    () { throw std::runtime_error("Band fluxes are not available"); }
        virtual Array_gpu<Float,3>& get_bnd_flux_dn    () { throw std::runtime_error("Band fluxes are not available"); }
        virtual Array_gpu<Float,3>& get_bnd_flux_dn_dir() { throw std::runtime_error("Band fluxes are not available"); }
        virtual Array_gpu<Float,3>& get_bnd_flux_net   () { throw std::runtime_error("Band fluxes are not available"); }
    private:
        Array_gpu<Float,2> flux_up, flux_dn, flux_dn_dir, flux_net;
};

class Fluxes_byband_gpu : public Fluxes_broadband_gpu
{
    public:
        Fluxes_byband_gpu(const int ncol, const int nlev, const int nbnd);
        virtual ~Fluxes_byband_gpu() {}
        virtual void reduce(const Array_gpu<Float,3>& gpt_flux_up, const Array_gpu<Float,3>& gpt_flux_dn,
                const std::unique_ptr<Optical_props_arry_gpu>& optical_props, const Bool top_at_1);
        virtual void reduce(const Array_gpu<Float,3>& gpt_flux_up, const Array_gpu<Float,3>& gpt_flux_dn,
                const Array_gpu<Float,3>& gpt_flux_dn_dir,
                const std::unique_ptr<Optical_props_arry_gpu>& optical_props, const Bool top_at_1);
        Array_gpu<Float,3>& get_bnd_flux_up    () { return bnd_flux_up;     }
        Array_gpu<Float,3>& get_bnd_flux_dn    () { return bnd_flux_dn;     }
        Array_gpu<Float,3>& get_bnd_flux_dn_dir() { return bnd_flux_dn_dir; }
        Array_gpu<Float,3>& get_bnd_flux_net   () { return bnd_flux_net;    }
    private:
        Array_gpu<Float,3> bnd_flux_up, bnd_flux_dn, bnd_flux_dn_dir, bnd_flux_net;
};
#endif
