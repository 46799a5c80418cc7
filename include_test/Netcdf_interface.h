/*
 * Netcdf_file / Netcdf_handle / Netcdf_variable<T> -- the subset of /root/reference/include_test/Netcdf_interface.h that the
 * hot-path drivers use (get_dimension_size, variable_exists, get_variable_dimensions, get_variable<T>, add_dimension,
 * add_variable<T>(...).insert), with the same call signatures, so load_and_init_gas_optics / solve_radiation read like
 * the reference.
 *
 * Storage backends, chosen per file by its first bytes:
 *  - NetCDF-4 (HDF5): the format of rrtmgp-data, aerosol_optics.nc and rte_rrtmgp_input.nc. NetCDF-C is absent from the build
 *    image; include_test/Netcdf_hdf5.h reads and writes the NetCDF-4 on-disk conventions with the HDF5 C library
 *    (dlopen'ed on first use). Output is NetCDF-4 when RRX_OUTPUT_FORMAT=netcdf4 (or set_output_format("netcdf4")).
 *  - "RRXB" containers (a flat, self-describing binary: named dimensions + named typed variables in C order, i.e. exactly the
 *    NetCDF data model the drivers rely on), which rte-rrtmgp-cpp_amd/rrxio.py reads and writes without any library: the
 *    format of the synthetic test inputs and the default output format.
 *
 * File layout (little endian):  "RRXB1\0\0\0" | u32 ndim | ndim x {u32 len, name, i64 size}
 *                               | u32 nvar | nvar x {u32 len, name, u8 dtype, u32 rank, rank x {u32 len, dimname}, i64 nbytes, data}
 * dtype: 0 = f64, 1 = f32, 2 = i32, 3 = i8 (char / Bool)
 */
#ifndef NETCDF_INTERFACE_H
#define NETCDF_INTERFACE_H
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>
#include "Netcdf_hdf5.h"
#include "Netcdf_classic.h"

enum class Netcdf_mode { Create, Read, Write };

namespace rrxb
{
    template<typename T> struct Dtype;
    template<> struct Dtype<double>      { static constexpr uint8_t id = 0; };
    template<> struct Dtype<float>       { static constexpr uint8_t id = 1; };
    template<> struct Dtype<int>         { static constexpr uint8_t id = 2; };
    template<> struct Dtype<signed char> { static constexpr uint8_t id = 3; };
    template<> struct Dtype<char>        { static constexpr uint8_t id = 3; };
    inline size_t elem_size(const uint8_t dtype) { return dtype == 0 ? 8 : (dtype == 3 ? 1 : 4); }

    template<typename T>
    std::vector<T> convert(const Var& v, const size_t n, const std::string& name)
    {
        const size_t have = v.bytes.size() / elem_size(v.dtype);
        if (have < n) throw std::runtime_error("variable " + name + " holds fewer values than requested");
        std::vector<T> out(n);
        for (size_t i=0; i<n; ++i)
        {
            switch (v.dtype)
            {
                case 0: { double x; std::memcpy(&x, &v.bytes[8*i], 8); out[i] = static_cast<T>(x); break; }
                case 1: { float x;  std::memcpy(&x, &v.bytes[4*i], 4); out[i] = static_cast<T>(x); break; }
                case 2: { int x;    std::memcpy(&x, &v.bytes[4*i], 4); out[i] = static_cast<T>(x); break; }
                default: out[i] = static_cast<T>(v.bytes[i]);
            }
        }
        return out;
    }
}

class Netcdf_handle;

template<typename T>
class Netcdf_variable
{
    public:
        Netcdf_variable(Netcdf_handle& nc, const std::string& name, const std::vector<int>& dim_sizes) :
            nc(nc), name(name), dim_sizes(dim_sizes) {}
        // whole-variable or hyperslab insert at `i_start` (only leading-dimension offsets are used by the drivers)
        void insert(const std::vector<T>& values, const std::vector<int> i_start);
        void insert(const T value, const std::vector<int> i_start) { insert(std::vector<T>{value}, i_start); }
    private:
        Netcdf_handle& nc;
        std::string name;
        std::vector<int> dim_sizes;
};

class Netcdf_handle
{
    public:
        virtual ~Netcdf_handle() {}

        void add_dimension(const std::string& dim_name, const int dim_size = 0)
        {
            if (dims.count(dim_name)) throw std::runtime_error("dimension " + dim_name + " exists");
            dims[dim_name] = dim_size; dim_order.push_back(dim_name); dirty = true;
        }

        template<typename T>
        Netcdf_variable<T> add_variable(const std::string& var_name, const std::vector<std::string> dim_names = {})
        {
            rrxb::Var v; v.dtype = rrxb::Dtype<T>::id; v.dims = dim_names;
            size_t n = 1; std::vector<int> sizes;
            for (const auto& d : dim_names)
            {
                if (!dims.count(d)) throw std::runtime_error("unknown dimension " + d);
                n *= dims[d]; sizes.push_back(int(dims[d]));
            }
            v.bytes.assign(n*sizeof(T), 0);
            vars[var_name] = std::move(v); var_order.push_back(var_name); dirty = true;
            return Netcdf_variable<T>(*this, var_name, sizes);
        }

        int get_dimension_size(const std::string& name) const
        {
            auto it = dims.find(name);
            if (it == dims.end()) throw std::runtime_error("dimension " + name + " not found");
            return int(it->second);
        }

        bool variable_exists(const std::string& name) const { return vars.count(name) != 0; }

        std::map<std::string, int> get_variable_dimensions(const std::string& name) const
        {
            std::map<std::string, int> out;
            for (const auto& d : var(name).dims) out[d] = get_dimension_size(d);
            return out;
        }

        template<typename T>
        std::vector<T> get_variable(const std::string& name, const std::vector<int>& i_count) const
        {
            const size_t n = std::accumulate(i_count.begin(), i_count.end(), size_t(1), std::multiplies<size_t>());
            return rrxb::convert<T>(var(name), n, name);
        }

        template<typename T>
        T get_variable(const std::string& name) const { return rrxb::convert<T>(var(name), 1, name)[0]; }

        template<typename T>
        void store(const std::string& name, const std::vector<T>& values, const size_t offset)
        {
            rrxb::Var& v = vars.at(name);
            if ((offset + values.size())*sizeof(T) > v.bytes.size()) throw std::runtime_error("insert beyond the end of " + name);
            std::memcpy(&v.bytes[offset*sizeof(T)], values.data(), values.size()*sizeof(T));
            dirty = true;
        }

        // every dimension and variable of another file (any backend), e.g. to convert NetCDF-4 <-> RRXB
        void copy_contents_of(const Netcdf_handle& o)
        {
            for (const auto& d : o.dim_order) if (!dims.count(d)) { dims[d] = o.dims.at(d); dim_order.push_back(d); }
            for (const auto& n : o.var_order)
            {
                const rrxb::Var& v = o.var(n);
                rrxb::Var c; c.dtype = v.dtype; c.dims = v.dims; c.bytes = v.bytes;
                if (!vars.count(n)) var_order.push_back(n);
                vars[n] = std::move(c);
            }
            dirty = true;
        }
        const std::vector<std::string>& variable_names() const { return var_order; }

    protected:
        const rrxb::Var& var(const std::string& name) const
        {
            auto it = vars.find(name);
            if (it == vars.end()) throw std::runtime_error("variable " + name + " not found");
            if (it->second.loader)                 // lazy backend (NetCDF-4): the data is read on first access
            {
                auto loader = std::move(it->second.loader);
                it->second.loader = nullptr;
                loader(it->second);
            }
            return it->second;
        }
        std::map<std::string, int64_t> dims;
        std::vector<std::string> dim_order;
        mutable std::map<std::string, rrxb::Var> vars;
        std::vector<std::string> var_order;
        bool dirty = false;
};

template<typename T>
void Netcdf_variable<T>::insert(const std::vector<T>& values, const std::vector<int> i_start)
{
    size_t offset = 0, stride = 1;
    for (int d=int(dim_sizes.size())-1; d>=0; --d)
    {
        if (d < int(i_start.size())) offset += size_t(i_start[d]) * stride;
        stride *= dim_sizes[d];
    }
    nc.store<T>(name, values, offset);
}

class Netcdf_file : public Netcdf_handle
{
    public:
        Netcdf_file(const std::string& name, Netcdf_mode mode) : file_name(name), mode(mode)
        {
            if (mode == Netcdf_mode::Read || mode == Netcdf_mode::Write) read();
        }
        ~Netcdf_file() { try { sync(); } catch (...) {} }

        // "rrxb" (default) or "netcdf4"; the environment variable RRX_OUTPUT_FORMAT sets the default of the process
        void set_output_format(const std::string& fmt) { netcdf4_out = (fmt == "netcdf4"); }

        void sync()
        {
            if (mode == Netcdf_mode::Read || !dirty) return;
            if (netcdf4_out)
            {
#ifdef RRX_HAVE_HDF5_HEADERS
                for (auto& kv : vars) (void)var(kv.first);          // anything still lazy is loaded before the file is replaced
                rrx_h5::write_file(file_name, dims, dim_order, vars, var_order);
                dirty = false;
                return;
#else
                throw std::runtime_error("NetCDF-4 output needs the HDF5 headers at build time");
#endif
            }
            std::ofstream f(file_name, std::ios::binary | std::ios::trunc);
            if (!f) throw std::runtime_error("cannot write " + file_name);
            auto put_u32 = [&](uint32_t v) { f.write(reinterpret_cast<char*>(&v), 4); };
            auto put_i64 = [&](int64_t v) { f.write(reinterpret_cast<char*>(&v), 8); };
            auto put_str = [&](const std::string& s) { put_u32(uint32_t(s.size())); f.write(s.data(), s.size()); };
            f.write("RRXB1\0\0\0", 8);
            put_u32(uint32_t(dim_order.size()));
            for (const auto& d : dim_order) { put_str(d); put_i64(dims[d]); }
            put_u32(uint32_t(var_order.size()));
            for (const auto& n : var_order)
            {
                const rrxb::Var& v = var(n);
                put_str(n); f.write(reinterpret_cast<const char*>(&v.dtype), 1);
                put_u32(uint32_t(v.dims.size()));
                for (const auto& d : v.dims) put_str(d);
                put_i64(int64_t(v.bytes.size()));
                f.write(v.bytes.data(), v.bytes.size());
            }
            dirty = false;
        }

    private:
        void read()
        {
            if (rrx_h5::is_hdf5(file_name))
            {
#ifdef RRX_HAVE_HDF5_HEADERS
                rrx_h5::read_file(file_name, dims, dim_order, vars, var_order);
                netcdf4_out = true;                                  // a file opened for writing keeps its format
                return;
#else
                throw std::runtime_error(file_name + " is a NetCDF-4 file: this build has no HDF5 support");
#endif
            }
            if (rrx_cdf::version(file_name) != 0)                       // classic NetCDF (CDF-1 / CDF-2), read-only
            {
                rrx_cdf::read_file(file_name, dims, dim_order, vars, var_order);
                return;
            }
            std::ifstream f(file_name, std::ios::binary);
            if (!f) throw std::runtime_error("cannot open " + file_name);
            char magic[8]; f.read(magic, 8);
            if (std::memcmp(magic, "RRXB1", 5) != 0) throw std::runtime_error(file_name + " is neither NetCDF (classic or NetCDF-4) nor an RRXB container");
            auto get_u32 = [&]() { uint32_t v; f.read(reinterpret_cast<char*>(&v), 4); return v; };
            auto get_i64 = [&]() { int64_t v; f.read(reinterpret_cast<char*>(&v), 8); return v; };
            auto get_str = [&]() { std::string s(get_u32(), '\0'); f.read(&s[0], s.size()); return s; };
            const uint32_t nd = get_u32();
            for (uint32_t i=0; i<nd; ++i) { std::string n = get_str(); dims[n] = get_i64(); dim_order.push_back(n); }
            const uint32_t nv = get_u32();
            for (uint32_t i=0; i<nv; ++i)
            {
                std::string n = get_str();
                rrxb::Var v; f.read(reinterpret_cast<char*>(&v.dtype), 1);
                const uint32_t rank = get_u32();
                for (uint32_t r=0; r<rank; ++r) v.dims.push_back(get_str());
                v.bytes.resize(size_t(get_i64()));
                f.read(v.bytes.data(), v.bytes.size());
                if (!f) throw std::runtime_error("truncated RRXB file " + file_name);
                vars[n] = std::move(v); var_order.push_back(n);
            }
        }
        std::string file_name;
        Netcdf_mode mode;
        bool netcdf4_out = []{ const char* e = std::getenv("RRX_OUTPUT_FORMAT"); return e != nullptr && std::string(e) == "netcdf4"; }();
};
#endif
