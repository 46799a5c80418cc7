// TEST INFRASTRUCTURE ONLY (build container only: needs /root/reference).
//
// Command-line runner around the reference's OWN plain-C++ CPU classes, compiled unmodified and in place by
// `make -C oracle refcpu`:  /root/reference/src/{Cloud_optics,Aerosol_optics,Optical_props,Gas_concs}.cpp.
// It builds the reference's Cloud_optics / Aerosol_optics objects from arrays read from a binary file, calls
//   Cloud_optics::cloud_optics(..., Optical_props_2str&) and (..., Optical_props_1scl&)   (src/Cloud_optics.cpp:111-232)
//   Aerosol_optics::aerosol_optics(Aerosol_concs&, rh, plev, Optical_props_2str&)          (src/Aerosol_optics.cpp:158-224)
// and writes the results. No header, library or generated code of the reference is replaced. The five rte_* Fortran
// symbols that src/Optical_props.cpp references (increment / delta_scale) are never reached on these two paths and are
// left unresolved at link time (an executable with lazy binding; they are NOT provided by the oracle).
//
// File format (all little endian): int32 n_int, int32 ints[n_int], then Float arrays back to back.
// Arrays are column-major with the first reference index fastest (= numpy C order with the axes reversed).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "Array.h"
#include "Gas_concs.h"
#include "Optical_props.h"
#include "Cloud_optics.h"
#include "Aerosol_optics.h"

namespace
{
    struct Reader
    {
        FILE* f;
        std::vector<int> ints;
        explicit Reader(const char* path)
        {
            f = std::fopen(path, "rb");
            if (!f) { std::perror(path); std::exit(2); }
            int n = 0;
            if (std::fread(&n, 4, 1, f) != 1) std::exit(2);
            ints.resize(n);
            if (n && std::fread(ints.data(), 4, n, f) != size_t(n)) std::exit(2);
        }
        std::vector<Float> floats(size_t n)
        {
            std::vector<Float> v(n);
            if (n && std::fread(v.data(), sizeof(Float), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
            return v;
        }
        template<int N> Array<Float,N> array(const std::array<int,N>& dims)
        {
            size_t n = 1;
            for (int d : dims) n *= d;
            return Array<Float,N>(floats(n), dims);
        }
        ~Reader() { std::fclose(f); }
    };

    void put(FILE* f, const std::vector<Float>& v) { std::fwrite(v.data(), sizeof(Float), v.size(), f); }

    int run_cloud(const char* in, const char* out)
    {
        Reader r(in);
        const int ncol = r.ints[0], nlay = r.ints[1], nbnd = r.ints[2], nliq = r.ints[3], nice = r.ints[4], nrgh = r.ints[5];
        std::vector<Float> s = r.floats(4);          // radliq_lwr, radliq_upr, diamice_lwr, diamice_upr
        auto band_lims_wvn = r.array<2>({2, nbnd});
        auto extliq = r.array<2>({nliq, nbnd}); auto ssaliq = r.array<2>({nliq, nbnd}); auto asyliq = r.array<2>({nliq, nbnd});
        auto extice = r.array<3>({nice, nbnd, nrgh}); auto ssaice = r.array<3>({nice, nbnd, nrgh}); auto asyice = r.array<3>({nice, nbnd, nrgh});
        auto clwp = r.array<2>({ncol, nlay}); auto ciwp = r.array<2>({ncol, nlay});
        auto reliq = r.array<2>({ncol, nlay}); auto deice = r.array<2>({ncol, nlay});

        Cloud_optics clouds(band_lims_wvn, s[0], s[1], Float(1.), s[2], s[3], Float(1.),
                            extliq, ssaliq, asyliq, extice, ssaice, asyice);
        Optical_props_2str p2(ncol, nlay, clouds);
        clouds.cloud_optics(clwp, ciwp, reliq, deice, p2);
        Optical_props_1scl p1(ncol, nlay, clouds);
        clouds.cloud_optics(clwp, ciwp, reliq, deice, p1);

        FILE* f = std::fopen(out, "wb");
        if (!f) { std::perror(out); return 2; }
        put(f, p2.get_tau().v()); put(f, p2.get_ssa().v()); put(f, p2.get_g().v()); put(f, p1.get_tau().v());
        std::fclose(f);
        return 0;
    }

    int run_aerosol(const char* in, const char* out)
    {
        Reader r(in);
        const int ncol = r.ints[0], nlay = r.ints[1], nbnd = r.ints[2], nhum = r.ints[3], nphobic = r.ints[4], nphilic = r.ints[5];
        auto band_lims_wvn = r.array<2>({2, nbnd});
        auto rh_upper = r.array<1>({nhum});
        auto mext_phobic = r.array<2>({nbnd, nphobic}); auto ssa_phobic = r.array<2>({nbnd, nphobic}); auto g_phobic = r.array<2>({nbnd, nphobic});
        auto mext_philic = r.array<3>({nbnd, nhum, nphilic}); auto ssa_philic = r.array<3>({nbnd, nhum, nphilic}); auto g_philic = r.array<3>({nbnd, nhum, nphilic});
        Aerosol_concs concs;
        for (int i=1; i<=11; ++i)
        {
            const std::string name = i < 10 ? "aermr0" + std::to_string(i) : "aermr" + std::to_string(i);
            const bool profile = r.ints[5 + i] != 0;          // given as a (1, nlay) profile, broadcast by the reference itself
            if (profile)
                concs.set_vmr(name, r.array<2>({1, nlay}));
            else
                concs.set_vmr(name, r.array<2>({ncol, nlay}));
        }
        auto rh = r.array<2>({ncol, nlay});
        auto plev = r.array<2>({ncol, nlay+1});

        Aerosol_optics aerosols(band_lims_wvn, rh_upper, mext_phobic, ssa_phobic, g_phobic, mext_philic, ssa_philic, g_philic);
        Optical_props_2str p2(ncol, nlay, aerosols);
        aerosols.aerosol_optics(concs, rh, plev, p2);

        FILE* f = std::fopen(out, "wb");
        if (!f) { std::perror(out); return 2; }
        put(f, p2.get_tau().v()); put(f, p2.get_ssa().v()); put(f, p2.get_g().v());
        std::fclose(f);
        return 0;
    }
}

int main(int argc, char** argv)
{
    if (argc != 4) { std::fprintf(stderr, "usage: %s cloud|aerosol in.bin out.bin\n", argv[0]); return 1; }
    try
    {
        if (!std::strcmp(argv[1], "cloud")) return run_cloud(argv[2], argv[3]);
        if (!std::strcmp(argv[1], "aerosol")) return run_aerosol(argv[2], argv[3]);
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "refcpu: %s\n", e.what());
        return 3;
    }
    return 1;
}
