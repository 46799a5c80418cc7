// Self-checks of the host classes that need a device but no input files; exported from librte_rrtmgp_hip.so for the GPU
// tests (tests/test_gpu_host_classes.py). Each returns 0 on success and a non-zero code naming the failed check.
#include <cmath>
#include <vector>
#include <cstdio>
#include <cstring>
#include <string>
#include "Array.h"
#include "Optical_props.h"
#include "Netcdf_interface.h"

namespace
{
    Optical_props_gpu two_bands()
    {
        Array<Float,2> wvn({2, 2});
        wvn({1, 1}) = 10.; wvn({2, 1}) = 250.; wvn({1, 2}) = 250.; wvn({2, 2}) = 500.;
        Array<int,2> lims({2, 2});
        lims({1, 1}) = 1; lims({2, 1}) = 3; lims({1, 2}) = 4; lims({2, 2}) = 6;
        return Optical_props_gpu(wvn, lims);
    }
}

extern "C"
{
// Optical_props_2str_gpu::delta_scale() on gas-only optical properties whose asymmetry is held in the lazy "g == 0"
// state (what the SW gas optics leave behind): the result must be the identity on tau and ssa, g must read back as
// zeros, and the stale contents of the g array must never reach the kernel (reference semantics with g = 0:
// /root/reference/src_kernels_cuda/optical_props_kernels.cu:140-161).
int rrx_host_selftest_delta_scale_gzero(void)
{
    try
    {
        const int ncol = 5, nlay = 4;
        Optical_props_2str_gpu op(ncol, nlay, two_bands());
        const int n = ncol*nlay*op.get_ngpt();
        Array<Float,3> tau({ncol, nlay, op.get_ngpt()}), ssa({ncol, nlay, op.get_ngpt()});
        for (int i=0; i<n; ++i) { tau.ptr()[i] = Float(0.01)*(i+1); ssa.ptr()[i] = Float(1.)/(i+2); }
        op.get_tau().set_data(tau); op.get_ssa().set_data(ssa);
        op.get_g().fill(Float(0.7));        // stale values of an earlier use of the workspace (e.g. cloud-incremented g)
        op.set_g_zero();
        if (op.get_g_or_null() != nullptr) return 1;
        op.delta_scale();
        Array<Float,3> t2(op.get_tau()), w2(op.get_ssa());
        for (int i=0; i<n; ++i)
            if (t2.ptr()[i] != tau.ptr()[i] || w2.ptr()[i] != ssa.ptr()[i]) return 2;
        if (op.get_g_or_null() != nullptr) return 3;               // still lazy: nothing materialised, nothing read
        Array<Float,3> g2(op.get_g());                              // first real access fills zeros
        for (int i=0; i<n; ++i) if (g2.ptr()[i] != Float(0.)) return 4;
        // and with a materialised g the kernel runs: f = g^2, tau' = tau (1 - ssa f)
        op.get_g().fill(Float(0.5));
        op.delta_scale();
        Array<Float,3> t3(op.get_tau());
        for (int i=0; i<n; ++i)
        {
            const Float want = tau.ptr()[i] * (Float(1.) - ssa.ptr()[i]*Float(0.25));
            if (std::abs(t3.ptr()[i] - want) > Float(1e-6)*std::abs(want)) return 5;
        }
        return 0;
    }
    catch (const std::exception&) { return 100; }
}

// Any supported file (NetCDF-4 or RRXB, recognised by its first bytes) rewritten in the other format: "rrxb" or "netcdf4".
int rrx_host_netcdf_convert(const char* in_path, const char* out_path, const char* format)
{
    try
    {
        Netcdf_file in(in_path, Netcdf_mode::Read);
        Netcdf_file out(out_path, Netcdf_mode::Create);
        out.set_output_format(format);
        out.copy_contents_of(in);
        out.sync();
        return 0;
    }
    catch (const std::exception& e) { std::fprintf(stderr, "rrx_host_netcdf_convert: %s\n", e.what()); return 1; }
}

// String attribute of a variable of a NetCDF-4 or classic NetCDF file (e.g. "units"); returns the length, -1 when absent or on error.
int rrx_host_netcdf_get_attr(const char* path, const char* var, const char* attr, char* buf, int buflen)
{
    if (rrx_cdf::version(path) != 0)                     // classic NetCDF
    {
        try
        {
            const std::string v = rrx_cdf::get_text_attr(path, var, attr);
            if (v.empty() || int(v.size()) >= buflen) return -1;
            std::memcpy(buf, v.c_str(), v.size() + 1);
            return int(v.size());
        }
        catch (const std::exception& e) { std::fprintf(stderr, "rrx_host_netcdf_get_attr: %s\n", e.what()); return -1; }
    }
#ifdef RRX_HAVE_HDF5_HEADERS
    try
    {
        const std::string v = rrx_h5::get_string_attr(path, var, attr);
        if (v.empty() || int(v.size()) >= buflen) return -1;
        std::memcpy(buf, v.c_str(), v.size() + 1);
        return int(v.size());
    }
    catch (const std::exception& e) { std::fprintf(stderr, "rrx_host_netcdf_get_attr: %s\n", e.what()); return -1; }
#else
    (void)path; (void)var; (void)attr; (void)buf; (void)buflen; return -1;
#endif
}

int rrx_host_netcdf_put_attr(const char* path, const char* var, const char* attr, const char* value)
{
#ifdef RRX_HAVE_HDF5_HEADERS
    try { rrx_h5::put_string_attr(path, var, attr, value); return 0; }
    catch (const std::exception& e) { std::fprintf(stderr, "rrx_host_netcdf_put_attr: %s\n", e.what()); return 1; }
#else
    (void)path; (void)var; (void)attr; (void)value; return 1;
#endif
}

// NetCDF-4 round trip through the HDF5 backend: dimensions, dimension order of each variable, f64 / f32 / i32 / char data,
// a scalar, a coordinate variable; no device involved.
int rrx_host_selftest_netcdf4(const char* dir)
{
    try
    {
        const std::string path = std::string(dir) + "/roundtrip.nc";
        std::vector<double> a(2*3*4); for (size_t i=0; i<a.size(); ++i) a[i] = 0.5*double(i) - 3.25;
        std::vector<float> b(4); for (size_t i=0; i<b.size(); ++i) b[i] = 1.5f*float(i);
        std::vector<int> c(3*2); for (size_t i=0; i<c.size(); ++i) c[i] = int(i)*7 - 5;
        const std::string names = "h2o     co2     o3      ";
        {
            Netcdf_file f(path, Netcdf_mode::Create);
            f.set_output_format("netcdf4");
            f.add_dimension("lay", 2); f.add_dimension("y", 3); f.add_dimension("x", 4); f.add_dimension("pair", 2);
            f.add_dimension("absorber", 3); f.add_dimension("string_len", 8);
            f.add_variable<double>("p_lay", {"lay", "y", "x"}).insert(a, {0, 0, 0});
            f.add_variable<float>("x", {"x"}).insert(b, {0});                                   // coordinate variable
            f.add_variable<int>("limits", {"y", "pair"}).insert(c, {0, 0});
            f.add_variable<char>("gas_names", {"absorber", "string_len"}).insert(std::vector<char>(names.begin(), names.end()), {0, 0});
            f.add_variable<double>("tsi_default").insert(1360.85, {});
        }
        FILE* fp = std::fopen(path.c_str(), "rb");
        unsigned char magic[4] = {0};
        if (!fp || std::fread(magic, 1, 4, fp) != 4 || magic[1] != 'H' || magic[2] != 'D' || magic[3] != 'F') return 1;
        std::fclose(fp);
        Netcdf_file r(path, Netcdf_mode::Read);
        if (r.get_dimension_size("lay") != 2 || r.get_dimension_size("y") != 3 || r.get_dimension_size("x") != 4 || r.get_dimension_size("string_len") != 8) return 2;
        const auto d = r.get_variable_dimensions("p_lay");
        if (d.size() != 3 || d.at("lay") != 2 || d.at("y") != 3 || d.at("x") != 4) return 3;
        if (r.get_variable<double>("p_lay", {2, 3, 4}) != a) return 4;
        if (r.get_variable<float>("x", {4}) != b) return 5;
        if (r.get_variable<int>("limits", {3, 2}) != c) return 6;
        const std::vector<char> g = r.get_variable<char>("gas_names", {3, 8});
        if (std::string(g.begin(), g.end()) != names) return 7;
        if (r.get_variable<double>("tsi_default") != 1360.85) return 8;
        if (!r.variable_exists("limits") || r.variable_exists("nope")) return 9;
        if (r.get_variable<float>("p_lay", {2, 3, 4})[5] != float(a[5])) return 10;              // type conversion on read
        return 0;
    }
    catch (const std::exception& e) { std::fprintf(stderr, "rrx_host_selftest_netcdf4: %s\n", e.what()); return 100; }
}

// The one real NetCDF-4 file of the reference tree (data/aerosol_optics.nc, written by netCDF4-python): dimensions and a few
// values, against numbers obtained independently with h5dump. Returns 0, or the index of the first check that failed.
int rrx_host_selftest_read_aerosol_file(const char* path)
{
    try
    {
        Netcdf_file f(path, Netcdf_mode::Read);
        if (f.get_dimension_size("band_sw") != 14 || f.get_dimension_size("relative_humidity") != 12 ||
            f.get_dimension_size("hydrophilic") != 7 || f.get_dimension_size("hydrophobic") != 14) return 1;
        const auto d = f.get_variable_dimensions("mass_ext_sw_hydrophilic");
        if (d.size() != 3 || d.at("hydrophilic") != 7 || d.at("relative_humidity") != 12 || d.at("band_sw") != 14) return 2;
        const std::vector<double> m = f.get_variable<double>("mass_ext_sw_hydrophilic", {7, 12, 14});
        const double want_m[14] = {123.197, 166.189, 433.198, 245.302, 317.993, 488.143, 809.627, 1028.66, 1760, 3071.97, 4759.77, 6852.52, 8921.85, 10958.8};
        for (int i=0; i<14; ++i) if (std::abs(m[(3*12 + 5)*14 + i] - want_m[i]) > 6e-6*want_m[i]) return 3;
        const std::vector<double> s = f.get_variable<double>("ssa_sw_hydrophobic", {14, 14});
        const double want_s[6] = {0.152659, 0.11512, 0.719504, 0.963146, 0.96859, 0.998038};
        for (int i=0; i<6; ++i) if (std::abs(s[13*14 + i] - want_s[i]) > 6e-6) return 4;
        const std::vector<double> rh = f.get_variable<double>("relative_humidity1", {12});
        const double want_rh[12] = {0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.85, 0.9, 0.95};
        for (int i=0; i<12; ++i) if (std::abs(rh[i] - want_rh[i]) > 1e-6) return 5;
        const std::vector<double> wn = f.get_variable<double>("wavenumber1_sw", {14});
        if (wn[0] != 820. || wn[1] != 2600. || wn[13] != 38000.) return 6;
        return 0;
    }
    catch (const std::exception& e) { std::fprintf(stderr, "rrx_host_selftest_read_aerosol_file: %s\n", e.what()); return 100; }
}
}
