/*
 * NetCDF-4 (HDF5) backend of Netcdf_file -- SURVEY 8(f1): reads the files the reference's drivers read through NetCDF-C
 * (/root/reference/include_test/Netcdf_interface.h: nc_open / nc_inq_dimid / nc_inq_dimlen / nc_inq_varid / nc_get_vara_*;
 * loaders /root/reference/src_test/Radiation_solver.cpp:70-329) -- rrtmgp-gas-{lw-g256,sw-g224}.nc, rrtmgp-clouds-*.nc,
 * aerosol_optics.nc, rte_rrtmgp_input.nc -- and writes rte_rrtmgp_output.nc as NetCDF-4, using the HDF5 C library directly
 * (NetCDF-C itself is not in the image; libhdf5 1.10 is, under /opt/conda). The library is dlopen'ed on first use, so
 * librte_rrtmgp_hip.so has no link-time dependency on it; RRX_HDF5_LIB names another libhdf5.so.
 *
 * NetCDF-4 on-disk conventions used (netCDF "NetCDF-4 File Format" specification):
 *   - a dimension is a dataset with attribute CLASS = "DIMENSION_SCALE"; a dimension without coordinate variable carries
 *     NAME = "This is a netCDF dimension but not a netCDF variable.<size>" and its extent is the dimension length;
 *   - a variable's dimensions are the attribute DIMENSION_LIST: one variable-length list of object references per axis;
 *   - NC_CHAR arrays are H5T_STRING of size 1; NC_DOUBLE / NC_FLOAT / NC_INT map to IEEE / two's-complement types.
 * Files written here carry the same attributes (+ _Netcdf4Dimid, _NCProperties) and are readable by netCDF tools.
 */
#ifndef NETCDF_HDF5_H
#define NETCDF_HDF5_H

#if defined(__has_include)
#  if __has_include(<hdf5.h>)
#    define RRX_HAVE_HDF5_HEADERS 1
#  endif
#endif

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace rrxb
{
    // in-memory model of one variable (shared with the RRXB backend): dtype 0 = f64, 1 = f32, 2 = i32, 3 = i8 (char / Bool)
    struct Var
    {
        uint8_t dtype = 0; std::vector<std::string> dims; std::vector<char> bytes;
        std::function<void(Var&)> loader;      // set by a lazy backend: fills `bytes` on first access
    };
}

#ifdef RRX_HAVE_HDF5_HEADERS
#include <dlfcn.h>
#include <hdf5.h>

namespace rrx_h5
{
    // ---- the slice of the HDF5 API in use, resolved from the dlopen'ed library
    struct Api
    {
        void* lib = nullptr;
#define RRX_H5_FN(name) decltype(&::name) name = nullptr;
        RRX_H5_FN(H5open) RRX_H5_FN(H5Eset_auto2)
        RRX_H5_FN(H5Fopen) RRX_H5_FN(H5Fcreate) RRX_H5_FN(H5Fclose)
        RRX_H5_FN(H5Literate) RRX_H5_FN(H5Lexists)
        RRX_H5_FN(H5Dopen2) RRX_H5_FN(H5Dcreate2) RRX_H5_FN(H5Dclose) RRX_H5_FN(H5Dget_space) RRX_H5_FN(H5Dget_type)
        RRX_H5_FN(H5Dread) RRX_H5_FN(H5Dwrite) RRX_H5_FN(H5Dvlen_reclaim)
        RRX_H5_FN(H5Screate_simple) RRX_H5_FN(H5Screate) RRX_H5_FN(H5Sclose)
        RRX_H5_FN(H5Sget_simple_extent_ndims) RRX_H5_FN(H5Sget_simple_extent_dims)
        RRX_H5_FN(H5Tget_class) RRX_H5_FN(H5Tget_size) RRX_H5_FN(H5Tcopy) RRX_H5_FN(H5Tset_size) RRX_H5_FN(H5Tclose)
        RRX_H5_FN(H5Tvlen_create) RRX_H5_FN(H5Tset_strpad) RRX_H5_FN(H5Tis_variable_str) RRX_H5_FN(H5free_memory)
        RRX_H5_FN(H5Aexists) RRX_H5_FN(H5Aopen) RRX_H5_FN(H5Aread) RRX_H5_FN(H5Aclose) RRX_H5_FN(H5Aget_type) RRX_H5_FN(H5Aget_space)
        RRX_H5_FN(H5Acreate2) RRX_H5_FN(H5Awrite)
        RRX_H5_FN(H5Rdereference2) RRX_H5_FN(H5Rcreate) RRX_H5_FN(H5Iget_name) RRX_H5_FN(H5Oclose)
#undef RRX_H5_FN
        hid_t t_double = -1, t_float = -1, t_int = -1, t_schar = -1, t_c_s1 = -1, t_ref_obj = -1, t_f64le = -1, t_f32le = -1, t_i32le = -1, t_i8le = -1;
    };

    inline Api& api()
    {
        static Api a;
        if (a.lib != nullptr) return a;
        std::vector<std::string> cand;
        if (const char* e = std::getenv("RRX_HDF5_LIB")) cand.push_back(e);
        for (const char* c : {"libhdf5.so", "/opt/conda/lib/libhdf5.so", "libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so"}) cand.push_back(c);
        void* lib = nullptr;
        for (const auto& c : cand) { lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) throw std::runtime_error("NetCDF-4 file: libhdf5.so not found (set RRX_HDF5_LIB)");
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) throw std::runtime_error(std::string("libhdf5: missing symbol ") + n); return p; };
#define RRX_H5_LOAD(name) a.name = reinterpret_cast<decltype(a.name)>(sym(#name));
        RRX_H5_LOAD(H5open) RRX_H5_LOAD(H5Eset_auto2)
        RRX_H5_LOAD(H5Fopen) RRX_H5_LOAD(H5Fcreate) RRX_H5_LOAD(H5Fclose)
        RRX_H5_LOAD(H5Literate) RRX_H5_LOAD(H5Lexists)
        RRX_H5_LOAD(H5Dopen2) RRX_H5_LOAD(H5Dcreate2) RRX_H5_LOAD(H5Dclose) RRX_H5_LOAD(H5Dget_space) RRX_H5_LOAD(H5Dget_type)
        RRX_H5_LOAD(H5Dread) RRX_H5_LOAD(H5Dwrite) RRX_H5_LOAD(H5Dvlen_reclaim)
        RRX_H5_LOAD(H5Screate_simple) RRX_H5_LOAD(H5Screate) RRX_H5_LOAD(H5Sclose)
        RRX_H5_LOAD(H5Sget_simple_extent_ndims) RRX_H5_LOAD(H5Sget_simple_extent_dims)
        RRX_H5_LOAD(H5Tget_class) RRX_H5_LOAD(H5Tget_size) RRX_H5_LOAD(H5Tcopy) RRX_H5_LOAD(H5Tset_size) RRX_H5_LOAD(H5Tclose)
        RRX_H5_LOAD(H5Tvlen_create) RRX_H5_LOAD(H5Tset_strpad) RRX_H5_LOAD(H5Tis_variable_str) RRX_H5_LOAD(H5free_memory)
        RRX_H5_LOAD(H5Aexists) RRX_H5_LOAD(H5Aopen) RRX_H5_LOAD(H5Aread) RRX_H5_LOAD(H5Aclose) RRX_H5_LOAD(H5Aget_type) RRX_H5_LOAD(H5Aget_space)
        RRX_H5_LOAD(H5Acreate2) RRX_H5_LOAD(H5Awrite)
        RRX_H5_LOAD(H5Rdereference2) RRX_H5_LOAD(H5Rcreate) RRX_H5_LOAD(H5Iget_name) RRX_H5_LOAD(H5Oclose)
#undef RRX_H5_LOAD
        if (a.H5open() < 0) throw std::runtime_error("H5open failed");
        a.H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);              // errors are reported through return codes / exceptions here
        auto gvar = [&](const char* n) { return *reinterpret_cast<hid_t*>(sym(n)); };       // predefined types are library globals
        a.t_double = gvar("H5T_NATIVE_DOUBLE_g"); a.t_float = gvar("H5T_NATIVE_FLOAT_g"); a.t_int = gvar("H5T_NATIVE_INT_g");
        a.t_schar = gvar("H5T_NATIVE_SCHAR_g"); a.t_c_s1 = gvar("H5T_C_S1_g"); a.t_ref_obj = gvar("H5T_STD_REF_OBJ_g");
        a.t_f64le = gvar("H5T_IEEE_F64LE_g"); a.t_f32le = gvar("H5T_IEEE_F32LE_g"); a.t_i32le = gvar("H5T_STD_I32LE_g"); a.t_i8le = gvar("H5T_STD_I8LE_g");
        a.lib = lib;
        return a;
    }

    inline bool is_hdf5(const std::string& path)
    {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        unsigned char m[8] = {0};
        const size_t n = std::fread(m, 1, 8, f);
        std::fclose(f);
        static const unsigned char sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        return n == 8 && std::memcmp(m, sig, 8) == 0;
    }

    inline std::string read_string_attr(Api& a, hid_t obj, const char* name)
    {
        if (a.H5Aexists(obj, name) <= 0) return std::string();
        const hid_t at = a.H5Aopen(obj, name, H5P_DEFAULT);
        if (at < 0) return std::string();
        const hid_t ft = a.H5Aget_type(at);
        std::string out;
        if (a.H5Tget_class(ft) == H5T_STRING && a.H5Tis_variable_str(ft) > 0)
        {
            // NC_STRING attribute (netCDF4-python writes these for non-ASCII or when asked to): one variable-length string
            char* p = nullptr;
            const hid_t mt = a.H5Tcopy(a.t_c_s1); a.H5Tset_size(mt, H5T_VARIABLE);
            if (a.H5Aread(at, mt, &p) >= 0 && p) { out = p; a.H5free_memory(p); }
            a.H5Tclose(mt);
        }
        else if (a.H5Tget_class(ft) == H5T_STRING)
        {
            const size_t n = a.H5Tget_size(ft);
            if (n > 0 && n < (1u << 20))
            {
                std::vector<char> buf(n + 1, 0);
                const hid_t mt = a.H5Tcopy(a.t_c_s1); a.H5Tset_size(mt, n);
                if (a.H5Aread(at, mt, buf.data()) >= 0) out.assign(buf.data(), strnlen(buf.data(), n));
                a.H5Tclose(mt);
            }
        }
        a.H5Tclose(ft); a.H5Aclose(at);
        return out;
    }

    // names of the dimension scales attached to a dataset (attribute DIMENSION_LIST), in axis order
    inline std::vector<std::string> dimension_list(Api& a, hid_t dset, const int rank)
    {
        std::vector<std::string> out;
        if (rank == 0 || a.H5Aexists(dset, "DIMENSION_LIST") <= 0) return out;
        const hid_t at = a.H5Aopen(dset, "DIMENSION_LIST", H5P_DEFAULT);
        const hid_t sp = a.H5Aget_space(at);
        const hid_t mt = a.H5Tvlen_create(a.t_ref_obj);
        std::vector<hvl_t> lists(rank);
        if (a.H5Aread(at, mt, lists.data()) >= 0)
        {
            for (int d=0; d<rank; ++d)
            {
                std::string name;
                if (lists[d].len >= 1)
                {
                    const hid_t obj = a.H5Rdereference2(dset, H5P_DEFAULT, H5R_OBJECT, lists[d].p);
                    if (obj >= 0)
                    {
                        char buf[512] = {0};
                        if (a.H5Iget_name(obj, buf, sizeof(buf)) > 0) { name = buf; const auto s = name.rfind('/'); if (s != std::string::npos) name = name.substr(s + 1); }
                        a.H5Oclose(obj);
                    }
                }
                out.push_back(name);
            }
            a.H5Dvlen_reclaim(mt, sp, H5P_DEFAULT, lists.data());
        }
        a.H5Tclose(mt); a.H5Sclose(sp); a.H5Aclose(at);
        return out;
    }

    struct Scan { Api* a; hid_t file; std::vector<std::string> names; };
    inline herr_t collect_name(hid_t, const char* name, const H5L_info_t*, void* op)
    {
        static_cast<Scan*>(op)->names.push_back(name);
        return 0;
    }

    // Reads the root group of a NetCDF-4 / HDF5 file into the in-memory model: dimensions and (lazily loaded) variables.
    inline void read_file(const std::string& path, std::map<std::string, int64_t>& dims, std::vector<std::string>& dim_order,
                          std::map<std::string, rrxb::Var>& vars, std::vector<std::string>& var_order)
    {
        Api& a = api();
        const hid_t file = a.H5Fopen(path.c_str(), 0u /* H5F_ACC_RDONLY */, H5P_DEFAULT);
        if (file < 0) throw std::runtime_error("cannot open HDF5 file " + path);
        Scan sc{&a, file, {}};
        hsize_t idx = 0;
        a.H5Literate(file, H5_INDEX_NAME, H5_ITER_INC, &idx, collect_name, &sc);
        int anon = 0;
        for (const std::string& name : sc.names)
        {
            const hid_t ds = a.H5Dopen2(file, name.c_str(), H5P_DEFAULT);
            if (ds < 0) continue;                                     // a group (NetCDF groups are not used by the drivers' files)
            const hid_t sp = a.H5Dget_space(ds);
            const int rank = a.H5Sget_simple_extent_ndims(sp);
            std::vector<hsize_t> ext(std::max(rank, 1), 1);
            if (rank > 0) a.H5Sget_simple_extent_dims(sp, ext.data(), nullptr);
            const std::string cls = read_string_attr(a, ds, "CLASS");
            const std::string nm = read_string_attr(a, ds, "NAME");
            const bool is_dim = (cls == "DIMENSION_SCALE");
            const bool pure_dim = is_dim && nm.compare(0, 52, "This is a netCDF dimension but not a netCDF variable") == 0;
            if (is_dim && rank == 1 && !dims.count(name)) { dims[name] = int64_t(ext[0]); dim_order.push_back(name); }
            if (!pure_dim)
            {
                rrxb::Var v;
                const hid_t ft = a.H5Dget_type(ds);
                const H5T_class_t tc = a.H5Tget_class(ft);
                const size_t ts = a.H5Tget_size(ft);
                size_t str_len = 0;
                if (tc == H5T_FLOAT) v.dtype = (ts == 8) ? 0 : 1;
                else if (tc == H5T_INTEGER) v.dtype = (ts == 1) ? 3 : 2;
                else if (tc == H5T_STRING) { v.dtype = 3; if (ts > 1) str_len = ts; }
                else { a.H5Tclose(ft); a.H5Sclose(sp); a.H5Dclose(ds); continue; }     // compound / opaque / references: not data of this path
                a.H5Tclose(ft);
                v.dims = is_dim ? std::vector<std::string>{name} : dimension_list(a, ds, rank);
                if (int(v.dims.size()) != rank) v.dims.clear();
                for (int d=0; d<rank; ++d)
                {
                    if (int(v.dims.size()) <= d) v.dims.push_back(std::string());
                    if (v.dims[d].empty())
                    {
                        v.dims[d] = "_dim" + std::to_string(anon++) + "_" + name;              // plain HDF5 dataset without scales
                        dims[v.dims[d]] = int64_t(ext[d]); dim_order.push_back(v.dims[d]);
                    }
                    else if (!dims.count(v.dims[d])) { dims[v.dims[d]] = int64_t(ext[d]); dim_order.push_back(v.dims[d]); }
                }
                if (str_len > 0)
                {
                    const std::string sd = "_strlen_" + name;
                    dims[sd] = int64_t(str_len); dim_order.push_back(sd); v.dims.push_back(sd);
                }
                size_t n = 1; for (int d=0; d<rank; ++d) n *= size_t(ext[d]);
                const uint8_t dtype = v.dtype;
                v.loader = [path, name, n, dtype, str_len](rrxb::Var& var)
                {
                    Api& a2 = api();
                    const hid_t f2 = a2.H5Fopen(path.c_str(), 0u /* H5F_ACC_RDONLY */, H5P_DEFAULT);
                    const hid_t d2 = a2.H5Dopen2(f2, name.c_str(), H5P_DEFAULT);
                    if (f2 < 0 || d2 < 0) throw std::runtime_error("cannot read variable " + name + " of " + path);
                    herr_t rc;
                    if (str_len > 0)
                    {
                        var.bytes.assign(n*str_len, 0);
                        const hid_t mt = a2.H5Tcopy(a2.t_c_s1); a2.H5Tset_size(mt, str_len);
                        rc = a2.H5Dread(d2, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, var.bytes.data());
                        a2.H5Tclose(mt);
                    }
                    else
                    {
                        const size_t es = dtype == 0 ? 8 : (dtype == 3 ? 1 : 4);
                        var.bytes.assign(n*es, 0);
                        hid_t mt = dtype == 0 ? a2.t_double : (dtype == 1 ? a2.t_float : (dtype == 2 ? a2.t_int : a2.t_schar));
                        hid_t own = -1;
                        if (dtype == 3)
                        {
                            // NC_CHAR is a size-1 string type: read it with its own file type (raw bytes)
                            const hid_t ft2 = a2.H5Dget_type(d2);
                            if (a2.H5Tget_class(ft2) == H5T_STRING) { own = ft2; mt = ft2; } else a2.H5Tclose(ft2);
                        }
                        rc = a2.H5Dread(d2, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, var.bytes.data());
                        if (own >= 0) a2.H5Tclose(own);
                    }
                    a2.H5Dclose(d2); a2.H5Fclose(f2);
                    if (rc < 0) throw std::runtime_error("H5Dread failed for " + name + " in " + path);
                };
                vars[name] = std::move(v); var_order.push_back(name);
            }
            a.H5Sclose(sp); a.H5Dclose(ds);
        }
        a.H5Fclose(file);
    }

    inline void write_string_attr(Api& a, hid_t obj, const char* name, const std::string& value)
    {
        const hid_t t = a.H5Tcopy(a.t_c_s1); a.H5Tset_size(t, value.size() + 1); a.H5Tset_strpad(t, H5T_STR_NULLTERM);
        const hid_t s = a.H5Screate(H5S_SCALAR);
        const hid_t at = a.H5Acreate2(obj, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
        if (at >= 0) { a.H5Awrite(at, t, value.c_str()); a.H5Aclose(at); }
        a.H5Sclose(s); a.H5Tclose(t);
    }

    // String attribute of a variable (or of the file: var empty) -- what the RFMIP acceptance script needs ("units" scale factors);
    // the class API of the reference has no attribute access, so this stays outside Netcdf_handle.
    inline std::string get_string_attr(const std::string& path, const std::string& var, const std::string& attr)
    {
        Api& a = api();
        const hid_t file = a.H5Fopen(path.c_str(), 0u /* H5F_ACC_RDONLY */, H5P_DEFAULT);
        if (file < 0) throw std::runtime_error("cannot open HDF5 file " + path);
        std::string out;
        if (var.empty()) out = read_string_attr(a, file, attr.c_str());
        else
        {
            const hid_t ds = a.H5Dopen2(file, var.c_str(), H5P_DEFAULT);
            if (ds < 0) { a.H5Fclose(file); throw std::runtime_error("variable " + var + " not found in " + path); }
            out = read_string_attr(a, ds, attr.c_str());
            a.H5Dclose(ds);
        }
        a.H5Fclose(file);
        return out;
    }

    inline void put_string_attr(const std::string& path, const std::string& var, const std::string& attr, const std::string& value);

    // Writes the in-memory model as a NetCDF-4 file: one dimension-scale dataset per dimension, one dataset per variable
    // with its DIMENSION_LIST.
    inline void write_file(const std::string& path, const std::map<std::string, int64_t>& dims, const std::vector<std::string>& dim_order,
                           const std::map<std::string, rrxb::Var>& vars, const std::vector<std::string>& var_order)
    {
        Api& a = api();
        const hid_t file = a.H5Fcreate(path.c_str(), 2u /* H5F_ACC_TRUNC */, H5P_DEFAULT, H5P_DEFAULT);
        if (file < 0) throw std::runtime_error("cannot create " + path);
        write_string_attr(a, file, "_NCProperties", "version=2,rte-rrtmgp-cpp_amd=1,hdf5=1.10");
        std::map<std::string, hobj_ref_t> dim_ref;
        int dimid = 0;
        auto file_type = [&a](const uint8_t dtype) { return dtype == 0 ? a.t_f64le : (dtype == 1 ? a.t_f32le : (dtype == 2 ? a.t_i32le : a.t_i8le)); };
        auto mem_type = [&a](const uint8_t dtype) { return dtype == 0 ? a.t_double : (dtype == 1 ? a.t_float : (dtype == 2 ? a.t_int : a.t_schar)); };
        for (const std::string& d : dim_order)
        {
            const hsize_t n = hsize_t(dims.at(d));
            // a coordinate variable (1-D variable named like its dimension) IS the dimension scale and carries the data
            const auto cv = vars.find(d);
            const bool coord = cv != vars.end() && cv->second.dims.size() == 1 && cv->second.dims[0] == d;
            if (cv != vars.end() && !coord) throw std::runtime_error("NetCDF-4: variable " + d + " is named like a dimension it does not span");
            const hid_t sp = a.H5Screate_simple(1, &n, nullptr);
            const hid_t ds = a.H5Dcreate2(file, d.c_str(), coord ? file_type(cv->second.dtype) : a.t_f32le, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            if (ds < 0) throw std::runtime_error("cannot create dimension " + d);
            write_string_attr(a, ds, "CLASS", "DIMENSION_SCALE");
            if (coord)
            {
                if (a.H5Dwrite(ds, mem_type(cv->second.dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, cv->second.bytes.data()) < 0) throw std::runtime_error("H5Dwrite failed for " + d);
                write_string_attr(a, ds, "NAME", d);
            }
            else
            {
                char nm[96]; std::snprintf(nm, sizeof(nm), "This is a netCDF dimension but not a netCDF variable.%10llu", static_cast<unsigned long long>(n));
                write_string_attr(a, ds, "NAME", nm);
            }
            {
                const hid_t s = a.H5Screate(H5S_SCALAR);
                const hid_t at = a.H5Acreate2(ds, "_Netcdf4Dimid", a.t_i32le, s, H5P_DEFAULT, H5P_DEFAULT);
                const int id = dimid++;
                if (at >= 0) { a.H5Awrite(at, a.t_int, &id); a.H5Aclose(at); }
                a.H5Sclose(s);
            }
            a.H5Dclose(ds); a.H5Sclose(sp);
            hobj_ref_t ref;
            if (a.H5Rcreate(&ref, file, d.c_str(), H5R_OBJECT, -1) < 0) throw std::runtime_error("H5Rcreate failed");
            dim_ref[d] = ref;
        }
        for (const std::string& name : var_order)
        {
            const rrxb::Var& v = vars.at(name);
            if (dims.count(name)) continue;                           // coordinate variable: written with its dimension
            std::vector<hsize_t> ext;
            for (const auto& d : v.dims) ext.push_back(hsize_t(dims.at(d)));
            const hid_t sp = ext.empty() ? a.H5Screate(H5S_SCALAR) : a.H5Screate_simple(int(ext.size()), ext.data(), nullptr);
            hid_t ft = file_type(v.dtype), mt = mem_type(v.dtype), nc_char = -1;
            if (v.dtype == 3)
            {
                // NC_CHAR, as netCDF-C stores it: a fixed string type of size 1 (so that nc_get_vara_text accepts the variable)
                nc_char = a.H5Tcopy(a.t_c_s1); a.H5Tset_size(nc_char, 1); a.H5Tset_strpad(nc_char, H5T_STR_NULLTERM);
                ft = nc_char; mt = nc_char;
            }
            const hid_t ds = a.H5Dcreate2(file, name.c_str(), ft, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            if (ds < 0) throw std::runtime_error("cannot create variable " + name);
            if (a.H5Dwrite(ds, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.bytes.data()) < 0) throw std::runtime_error("H5Dwrite failed for " + name);
            if (!ext.empty())
            {
                std::vector<hvl_t> lists(ext.size());
                std::vector<hobj_ref_t> refs(ext.size());
                for (size_t d=0; d<ext.size(); ++d) { refs[d] = dim_ref.at(v.dims[d]); lists[d].len = 1; lists[d].p = &refs[d]; }
                const hid_t vt = a.H5Tvlen_create(a.t_ref_obj);
                const hsize_t nd = ext.size();
                const hid_t as = a.H5Screate_simple(1, &nd, nullptr);
                const hid_t at = a.H5Acreate2(ds, "DIMENSION_LIST", vt, as, H5P_DEFAULT, H5P_DEFAULT);
                if (at >= 0) { a.H5Awrite(at, vt, lists.data()); a.H5Aclose(at); }
                a.H5Sclose(as); a.H5Tclose(vt);
            }
            if (nc_char >= 0) a.H5Tclose(nc_char);
            a.H5Dclose(ds); a.H5Sclose(sp);
        }
        a.H5Fclose(file);
    }

    inline void put_string_attr(const std::string& path, const std::string& var, const std::string& attr, const std::string& value)
    {
        Api& a = api();
        const hid_t file = a.H5Fopen(path.c_str(), 1u /* H5F_ACC_RDWR */, H5P_DEFAULT);
        if (file < 0) throw std::runtime_error("cannot open HDF5 file " + path + " for writing");
        const hid_t ds = a.H5Dopen2(file, var.c_str(), H5P_DEFAULT);
        if (ds < 0) { a.H5Fclose(file); throw std::runtime_error("variable " + var + " not found in " + path); }
        write_string_attr(a, ds, attr.c_str(), value);
        a.H5Dclose(ds); a.H5Fclose(file);
    }
}
#else
namespace rrx_h5
{
    inline bool is_hdf5(const std::string&) { return false; }
}
#endif
#endif
