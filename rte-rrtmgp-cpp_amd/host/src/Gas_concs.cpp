// Gas_concs / Gas_concs_gpu: behaviour of /root/reference/src/Gas_concs.cpp and src_cuda/Gas_concs.cu.
#include "Gas_concs.h"

Gas_concs::Gas_concs(const Gas_concs& ref, const int start, const int size)
{
    const int end = start + size - 1;
    for (const auto& g : ref.gas_concs_map)
    {
        if (g.second.dim(1) == 1)                                   // scalar or profile: shared by all columns
            gas_concs_map.emplace(g.first, g.second);
        else
            gas_concs_map.emplace(g.first, g.second.subset({{ {start, end}, {1, g.second.dim(2)} }}));
    }
}

void Gas_concs::set_vmr(const std::string& name, const Float data)
{
    Array<Float,2> a({1, 1}); a({1, 1}) = data;
    gas_concs_map.erase(name); gas_concs_map.emplace(name, std::move(a));
}

void Gas_concs::set_vmr(const std::string& name, const Array<Float,1>& data)
{
    Array<Float,2> a(data.v(), {1, data.dim(1)});
    gas_concs_map.erase(name); gas_concs_map.emplace(name, std::move(a));
}

void Gas_concs::set_vmr(const std::string& name, const Array<Float,2>& data)
{
    gas_concs_map.erase(name); gas_concs_map.emplace(name, data);
}

const Array<Float,2>& Gas_concs::get_vmr(const std::string& name) const
{
    auto it = gas_concs_map.find(name);
    if (it == gas_concs_map.end()) throw std::runtime_error("Gas_concs: gas \"" + name + "\" is not available");
    return it->second;
}

Bool Gas_concs::exists(const std::string& name) const { return gas_concs_map.count(name) != 0; }


Gas_concs_gpu::Gas_concs_gpu(const Gas_concs& ref)
{
    for (const auto& g : ref.gas_concs_map)
        gas_concs_map.emplace(g.first, Array_gpu<Float,2>(g.second));
}

Gas_concs_gpu::Gas_concs_gpu(const Gas_concs_gpu& ref, const int start, const int size)
{
    const int end = start + size - 1;
    for (const auto& g : ref.gas_concs_map)
    {
        if (g.second.dim(1) == 1)
            gas_concs_map.emplace(g.first, g.second);
        else
            gas_concs_map.emplace(g.first, g.second.subset({{ {start, end}, {1, g.second.dim(2)} }}));
    }
}

const Array_gpu<Float,2>& Gas_concs_gpu::get_vmr(const std::string& name) const
{
    auto it = gas_concs_map.find(name);
    if (it == gas_concs_map.end()) throw std::runtime_error("Gas_concs_gpu: gas \"" + name + "\" is not available");
    return it->second;
}

void Gas_concs_gpu::set_vmr(const std::string& name, const Array<Float,2>& data)
{
    gas_concs_map.erase(name); gas_concs_map.emplace(name, Array_gpu<Float,2>(data));
}

void Gas_concs_gpu::set_vmr(const std::string& name, const Array_gpu<Float,2>& data)
{
    gas_concs_map.erase(name); gas_concs_map.emplace(name, data);
}

Bool Gas_concs_gpu::exists(const std::string& name) const { return gas_concs_map.count(name) != 0; }

Gas_concs_gpu Gas_concs_gpu::gathered(const Array_gpu<int,1>& perm, const int n_col, const int n_out) const
{
    Gas_concs_gpu out;
    for (const auto& g : gas_concs_map)
    {
        if (g.second.dim(1) != n_col || n_col == 1) { out.gas_concs_map.emplace(g.first, g.second); continue; }
        Array_gpu<Float,2> a({n_out, g.second.dim(2)});
        RRX_CALL(rrx_gather_cols, n_out, (unsigned long long)g.second.dim(2), perm.ptr(), n_col, g.second.ptr(), a.ptr());
        out.gas_concs_map.emplace(g.first, std::move(a));
    }
    return out;
}
