/* Gas_concs / Gas_concs_gpu -- interface of /root/reference/include/Gas_concs.h:33-92 (vmr as scalar (1,1),
 * profile (1,nlay) or field (ncol,nlay); subset constructor takes a 1-based column start and a size). */
#ifndef GAS_CONCS_H
#define GAS_CONCS_H
#include <map>
#include <string>
#include "Array.h"

class Gas_concs
{
    public:
        Gas_concs() = default;
        Gas_concs(const Gas_concs& gas_concs_ref, const int start, const int size);
        void set_vmr(const std::string& name, const Float data);
        void set_vmr(const std::string& name, const Array<Float,1>& data);
        void set_vmr(const std::string& name, const Array<Float,2>& data);
        const Array<Float,2>& get_vmr(const std::string& name) const;
        Bool exists(const std::string& name) const;
    private:
        std::map<std::string, Array<Float,2>> gas_concs_map;
        friend class Gas_concs_gpu;
};

class Gas_concs_gpu
{
    public:
        Gas_concs_gpu() = default;
        Gas_concs_gpu(const Gas_concs& gas_concs_ref);
        Gas_concs_gpu(const Gas_concs_gpu& gas_concs_ref, const int start, const int size);
        const Array_gpu<Float,2>& get_vmr(const std::string& name) const;
        void set_vmr(const std::string& name, const Array<Float,2>& data);
        void set_vmr(const std::string& name, const Array_gpu<Float,2>& data);
        Bool exists(const std::string& name) const;
        // (not in the reference) the same gases with their per-column fields gathered through a column index of n_out entries
        // (rrx_gather_cols: columns in another order and / or padded); scalars and profiles are shared as they are
        Gas_concs_gpu gathered(const Array_gpu<int,1>& perm, const int n_col, const int n_out) const;
    private:
        std::map<std::string, Array_gpu<Float,2>> gas_concs_map;
};
#endif
