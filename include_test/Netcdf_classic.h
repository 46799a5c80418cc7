/*
 * Netcdf_classic.h -- reader of the classic NetCDF formats (CDF-1 "classic" and CDF-2 "64-bit offset") for Netcdf_file.
 *
 * Some published radiation data sets (older RFMIP reference files among them) are still written in the pre-HDF5 format; it is a
 * plain big-endian layout -- header (dimensions, attributes, variables with their file offsets) followed by the fixed-size
 * variables and then the record variables interleaved per record -- so it is read here directly, without any library.
 * Supported: types byte, char, short (widened to int), int, float, double; fixed and record (unlimited first dimension)
 * variables. Not supported: CDF-5. Variables load lazily, like the NetCDF-4 backend's.
 */
#ifndef NETCDF_CLASSIC_H
#define NETCDF_CLASSIC_H
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "Netcdf_hdf5.h"        // rrxb::Var

namespace rrx_cdf
{
    inline int version(const std::string& path)          // 0 = not a classic NetCDF file
    {
        std::ifstream f(path, std::ios::binary);
        char m[4] = {0, 0, 0, 0};
        if (!f.read(m, 4)) return 0;
        return (m[0] == 'C' && m[1] == 'D' && m[2] == 'F' && (m[3] == 1 || m[3] == 2)) ? int(m[3]) : 0;
    }

    struct Reader
    {
        std::ifstream f; int ver;
        uint32_t u32() { unsigned char b[4]; if (!f.read(reinterpret_cast<char*>(b), 4)) throw std::runtime_error("truncated NetCDF header"); return (uint32_t(b[0]) << 24) | (uint32_t(b[1]) << 16) | (uint32_t(b[2]) << 8) | b[3]; }
        uint64_t u64() { const uint64_t hi = u32(); return (hi << 32) | u32(); }
        // names are bounded (NC_MAX_NAME = 256 in the classic format; header fields of a corrupt file must not turn into allocations)
        std::string name()
        {
            const uint32_t n = u32();
            if (n > 4096) throw std::runtime_error("NetCDF classic: implausible name length in the header (corrupt file?)");
            std::string s(n, '\0');
            if (n && !f.read(&s[0], n)) throw std::runtime_error("truncated NetCDF header");
            f.seekg((4 - n % 4) % 4, std::ios::cur);
            return s;
        }
    };

    inline size_t type_size(const uint32_t t)
    {
        switch (t) { case 1: case 2: return 1; case 3: return 2; case 4: case 5: return 4; case 6: return 8; }
        throw std::runtime_error("NetCDF classic: unsupported external type");
    }

    inline void skip_attributes(Reader& r)
    {
        const uint32_t tag = r.u32(), n = r.u32();
        if (tag == 0 && n == 0) return;
        if (tag != 0x0C) throw std::runtime_error("NetCDF classic: attribute list expected");
        for (uint32_t i=0; i<n; ++i)
        {
            r.name();
            const uint32_t t = r.u32(), ne = r.u32();
            const size_t bytes = size_t(ne)*type_size(t);
            r.f.seekg(std::streamoff((bytes + 3) / 4 * 4), std::ios::cur);
        }
    }

    // attribute list at the reader's position: returns the value of text attribute `want` (empty if absent), skipping the rest
    inline std::string find_text_attribute(Reader& r, const std::string& want)
    {
        std::string found;
        const uint32_t tag = r.u32(), n = r.u32();
        if (tag == 0 && n == 0) return found;
        if (tag != 0x0C) throw std::runtime_error("NetCDF classic: attribute list expected");
        for (uint32_t i=0; i<n; ++i)
        {
            const std::string nm = r.name();
            const uint32_t t = r.u32(), ne = r.u32();
            const size_t bytes = size_t(ne)*type_size(t), padded = (bytes + 3) / 4 * 4;
            if (t == 2 && nm == want) { std::string v(bytes, '\0'); r.f.read(&v[0], bytes); r.f.seekg(std::streamoff(padded - bytes), std::ios::cur); found = v.c_str(); }
            else r.f.seekg(std::streamoff(padded), std::ios::cur);
        }
        return found;
    }

    // text attribute of a variable (var empty: global attribute); empty string when absent
    inline std::string get_text_attr(const std::string& path, const std::string& var, const std::string& attr)
    {
        Reader r{std::ifstream(path, std::ios::binary), version(path)};
        if (r.ver == 0) throw std::runtime_error(path + " is not a classic NetCDF file");
        r.f.seekg(8);
        {
            const uint32_t tag = r.u32(), n = r.u32();
            if (tag == 0x0A) for (uint32_t i=0; i<n; ++i) { r.name(); r.u32(); }
        }
        const std::string global = find_text_attribute(r, attr);
        if (var.empty()) return global;
        const uint32_t tag = r.u32(), n = r.u32();
        if (tag != 0x0B) return std::string();
        for (uint32_t i=0; i<n; ++i)
        {
            const std::string nm = r.name();
            const uint32_t rank = r.u32();
            for (uint32_t d=0; d<rank; ++d) r.u32();
            const std::string v = find_text_attribute(r, attr);
            r.u32(); r.u32(); if (r.ver == 1) r.u32(); else r.u64();
            if (nm == var) return v;
        }
        throw std::runtime_error("variable " + var + " not found in " + path);
    }

    inline void read_file(const std::string& path, std::map<std::string, int64_t>& dims, std::vector<std::string>& dim_order,
                          std::map<std::string, rrxb::Var>& vars, std::vector<std::string>& var_order)
    {
        Reader r{std::ifstream(path, std::ios::binary), version(path)};
        if (r.ver == 0) throw std::runtime_error(path + " is not a classic NetCDF file");
        r.f.seekg(4);
        const uint32_t numrecs = r.u32();
        if (numrecs == 0xFFFFFFFFu) throw std::runtime_error("NetCDF classic: " + path + " was left in streaming mode (numrecs = STREAMING): record count unknown");
        std::vector<std::string> dim_names; std::vector<int64_t> dim_len;
        int rec_dim = -1;                                      // the record (unlimited) dimension has length 0 in the header
        {
            const uint32_t tag = r.u32(), n = r.u32();
            if (!(tag == 0 && n == 0))
            {
                if (tag != 0x0A) throw std::runtime_error("NetCDF classic: dimension list expected");
                for (uint32_t i=0; i<n; ++i)
                {
                    std::string nm = r.name();
                    const uint32_t len = r.u32();
                    if (len == 0) rec_dim = int(i);
                    dim_names.push_back(nm); dim_len.push_back(len == 0 ? int64_t(numrecs) : int64_t(len));
                    dims[nm] = dim_len.back(); dim_order.push_back(nm);
                }
            }
        }
        skip_attributes(r);                                    // global attributes
        struct VarInfo { std::string name; std::vector<int> dimids; uint32_t type; uint64_t vsize, begin; bool record; };
        std::vector<VarInfo> infos;
        {
            const uint32_t tag = r.u32(), n = r.u32();
            if (!(tag == 0 && n == 0))
            {
                if (tag != 0x0B) throw std::runtime_error("NetCDF classic: variable list expected");
                for (uint32_t i=0; i<n; ++i)
                {
                    VarInfo v; v.name = r.name();
                    const uint32_t rank = r.u32();
                    if (rank > 1024) throw std::runtime_error("NetCDF classic: implausible rank of variable " + v.name);
                    for (uint32_t d=0; d<rank; ++d)
                    {
                        const uint32_t id = r.u32();
                        if (id >= dim_names.size()) throw std::runtime_error("NetCDF classic: variable " + v.name + " refers to a dimension that does not exist");
                        v.dimids.push_back(int(id));
                    }
                    skip_attributes(r);
                    v.type = r.u32(); v.vsize = r.u32(); v.begin = (r.ver == 1) ? uint64_t(r.u32()) : r.u64();
                    v.record = false;
                    infos.push_back(v);
                }
            }
        }
        r.f.seekg(0, std::ios::end);
        const uint64_t file_size = uint64_t(r.f.tellg());
        for (const auto& v : infos)
            if (v.begin > file_size) throw std::runtime_error("NetCDF classic: data of variable " + v.name + " starts beyond the end of " + path);
        uint64_t recsize = 0;
        int nrecvars = 0;
        for (auto& v : infos)
            if (!v.dimids.empty() && v.dimids[0] == rec_dim) { v.record = true; recsize += v.vsize; ++nrecvars; }
        for (const auto& v : infos)
        {
            rrxb::Var out;
            out.dtype = (v.type == 6) ? 0 : (v.type == 5) ? 1 : (v.type == 3 || v.type == 4) ? 2 : 3;
            size_t n = 1, per_rec = 1;
            for (size_t d=0; d<v.dimids.size(); ++d)
            {
                out.dims.push_back(dim_names.at(v.dimids[d]));
                n *= size_t(dim_len.at(v.dimids[d]));
                if (d > 0 || !v.record) per_rec *= size_t(dim_len.at(v.dimids[d]));
            }
            const uint32_t type = v.type; const uint64_t begin = v.begin; const bool record = v.record;
            // a single record variable is stored without the padding between records (format quirk)
            const uint64_t stride = (record && nrecvars == 1) ? uint64_t(per_rec)*type_size(type) : recsize;
            const size_t nrec = record ? size_t(numrecs) : 1;
            // the header's sizes must fit the file before anything is allocated for them
            if (uint64_t(n)*type_size(type) > file_size) throw std::runtime_error("NetCDF classic: variable " + v.name + " is larger than " + path);
            out.loader = [path, type, begin, record, stride, nrec, per_rec, n](rrxb::Var& var)
            {
                std::ifstream g(path, std::ios::binary);
                const size_t es = type_size(type), chunk = record ? per_rec : n;
                std::vector<unsigned char> raw(n*es);
                for (size_t irec=0; irec<nrec; ++irec)
                {
                    g.seekg(std::streamoff(begin + irec*stride));
                    if (!g.read(reinterpret_cast<char*>(raw.data() + irec*chunk*es), std::streamsize(chunk*es)))
                        throw std::runtime_error("NetCDF classic: truncated data in " + path);
                }
                const size_t os = (type == 6) ? 8 : (type == 1 || type == 2) ? 1 : 4;
                var.bytes.assign(n*os, 0);
                for (size_t i=0; i<n; ++i)
                {
                    const unsigned char* p = raw.data() + i*es;
                    char* o = var.bytes.data() + i*os;
                    if (es == 1) o[0] = char(p[0]);
                    else if (es == 2) { const int32_t x = int16_t(uint16_t(p[0]) << 8 | p[1]); std::memcpy(o, &x, 4); }
                    else if (es == 4) { const uint32_t x = (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; std::memcpy(o, &x, 4); }
                    else { uint64_t x = 0; for (int b=0; b<8; ++b) x = (x << 8) | p[b]; std::memcpy(o, &x, 8); }
                }
            };
            vars[v.name] = std::move(out); var_order.push_back(v.name);
        }
    }
}
#endif
