// test_rte_rrtmgp_gpu -- stand-alone driver with the command line, file names, variable names and flow of the reference's
// /root/reference/src_test/test_rte_rrtmgp.cu:196-818: read rte_rrtmgp_input, build the solvers from
// coefficients_{lw,sw} (+ cloud_coefficients_*), upload, solve on the GPU (1 warm-up + 1 timed run, 10 more with
// --timings), download, write rte_rrtmgp_output. Files are RRXB containers (include_test/Netcdf_interface.h), extension .nc
// kept so that run scripts need no change; two extra options: --broadband-solvers (on by default: the solvers sum the g-points
// themselves; --no-broadband-solvers restores per-g-point fluxes + sum_broadband) and the environment variable RRX_COL_BLOCK (columns per block, default 16384).
// --ngpus=N: the process becomes the launcher of N ranks of itself, one per GPU; rank r solves the contiguous column
// range rrx_column_range(r, N, ncol), the broadband (and optional band / optical) outputs are all-gathered over RCCL
// (include/rrx_rccl.h, librrx_rccl.so loaded on demand) and rank 0 writes the output file.
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <csignal>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>
#include <iomanip>
#include <iostream>
#include <map>
#include <numeric>
#include <sstream>

#include "Status.h"
#include "Netcdf_interface.h"
#include "Array.h"
#include "Gas_concs.h"
#include "Radiation_solver.h"
#include "rrx_rccl.h"

extern char** environ;

namespace
{
    // ---- multi-GPU context of one rank: the RCCL library is loaded only when there is more than one rank
    struct Ranks
    {
        int world = 1, rank = 0;
        void* lib = nullptr;
        void* comm = nullptr;
        decltype(&rrx_comm_get_unique_id) get_unique_id = nullptr;
        decltype(&rrx_comm_id_to_file) id_to_file = nullptr;
        decltype(&rrx_comm_id_from_file) id_from_file = nullptr;
        decltype(&rrx_comm_create) create = nullptr;
        decltype(&rrx_comm_destroy) destroy = nullptr;
        decltype(&rrx_rccl_last_error) last_error = nullptr;
#ifdef RTE_USE_SP
        decltype(&rrx_allgather_fluxes_f32) allgather = nullptr;
#else
        decltype(&rrx_allgather_fluxes_f64) allgather = nullptr;
#endif
        int col_s = 0, col_e = 0;            // this rank's columns [col_s, col_e), 0-based

        static std::string library_dir()
        {
            Dl_info info;
            if (dladdr(reinterpret_cast<void*>(&library_dir), &info) == 0 || !info.dli_fname) return ".";
            const std::string p(info.dli_fname);
            const size_t k = p.rfind('/');
            return k == std::string::npos ? "." : p.substr(0, k);
        }

        void check(const int rc, const char* what) const
        {
            if (rc != 0) throw std::runtime_error(std::string(what) + ": " + (last_error ? last_error() : "?"));
        }

        void init(const int world_, const int rank_, const int n_col_total)
        {
            world = world_; rank = rank_;
            col_s = 0; col_e = n_col_total;
            if (world == 1) return;
            if (n_col_total < world) throw std::runtime_error("fewer columns than GPUs");
            const std::string path = library_dir() + "/librrx_rccl.so";
            lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!lib) throw std::runtime_error("cannot load " + path + ": " + dlerror());
            auto sym = [&](const char* n) { void* q = dlsym(lib, n); if (!q) throw std::runtime_error(std::string("librrx_rccl: missing ") + n); return q; };
            get_unique_id = reinterpret_cast<decltype(get_unique_id)>(sym("rrx_comm_get_unique_id"));
            id_to_file = reinterpret_cast<decltype(id_to_file)>(sym("rrx_comm_id_to_file"));
            id_from_file = reinterpret_cast<decltype(id_from_file)>(sym("rrx_comm_id_from_file"));
            create = reinterpret_cast<decltype(create)>(sym("rrx_comm_create"));
            destroy = reinterpret_cast<decltype(destroy)>(sym("rrx_comm_destroy"));
            last_error = reinterpret_cast<decltype(last_error)>(sym("rrx_rccl_last_error"));
#ifdef RTE_USE_SP
            allgather = reinterpret_cast<decltype(allgather)>(sym("rrx_allgather_fluxes_f32"));
#else
            allgather = reinterpret_cast<decltype(allgather)>(sym("rrx_allgather_fluxes_f64"));
#endif
            reinterpret_cast<decltype(&rrx_column_range)>(sym("rrx_column_range"))(rank, world, n_col_total, &col_s, &col_e);
            int n_dev = 0;
            rrx_host::check(rrx_device_count(&n_dev));
            if (n_dev < 1) throw std::runtime_error("no GPU visible");
            rrx_host::check(rrx_set_device(rank % n_dev));
            const char* id_file = std::getenv("RRX_COMM_FILE");
            if (!id_file) throw std::runtime_error("RRX_COMM_FILE is not set (ranks are started by --ngpus)");
            char id[RRX_COMM_ID_BYTES];
            if (rank == 0) { check(get_unique_id(id), "rrx_comm_get_unique_id"); check(id_to_file(id_file, id), "rrx_comm_id_to_file"); }
            else check(id_from_file(id_file, id, 120), "rrx_comm_id_from_file");
            check(create(world, rank, id, &comm), "rrx_comm_create");
        }

        // (n_local, rest...) with the column fastest -> (n_total, rest...) on every rank
        template<int N>
        Array_gpu<Float,N> gather(const Array_gpu<Float,N>& local, const int n_col_total) const
        {
            if (world == 1) return local;
            std::array<int,N> d;
            int nrows = 1;
            for (int i=1; i<=N; ++i) { d[i-1] = local.dim(i); if (i > 1) nrows *= local.dim(i); }
            d[0] = n_col_total;
            Array_gpu<Float,N> full(d);
            const int n_max = (n_col_total + world - 1) / world;
            Array_gpu<Float,1> scratch({(world + 1) * nrows * n_max});
            check(allgather(comm, nrows, n_col_total, local.ptr(), full.ptr(), scratch.ptr(), nullptr), "rrx_allgather_fluxes");
            rrx_host::check(rrx_synchronize(nullptr));
            return full;
        }

        ~Ranks() { if (comm && destroy) destroy(comm); }
    };

    // `--ngpus=N` / `--ngpus N` is taken out of the argument list (the other options are on/off switches)
    int extract_ngpus(std::vector<std::string>& args)
    {
        int n = 1;
        for (size_t i=0; i<args.size(); )
        {
            if (args[i].compare(0, 8, "--ngpus=") == 0) { n = std::atoi(args[i].c_str() + 8); args.erase(args.begin() + i); }
            else if (args[i] == "--ngpus" && i + 1 < args.size()) { n = std::atoi(args[i+1].c_str()); args.erase(args.begin() + i, args.begin() + i + 2); }
            else ++i;
        }
        if (n < 1) throw std::runtime_error("--ngpus needs a positive number");
        return n;
    }

    // Launcher: N children of the stand-alone driver binary, nothing here touches the GPU. Returns the worst exit status.
    int launch_ranks(const int n, const std::vector<std::string>& args)
    {
        char self[4096];
        const ssize_t len = readlink("/proc/self/exe", self, sizeof(self) - 1);
        std::string exe = len > 0 ? std::string(self, len) : std::string();
        if (const char* e = std::getenv("RRX_DRIVER_EXE")) exe = e;
        else if (exe.find("test_rte_rrtmgp_gpu") == std::string::npos) exe = Ranks::library_dir() + "/test_rte_rrtmgp_gpu";   // called through the library
        if (access(exe.c_str(), X_OK) != 0) throw std::runtime_error("multi-GPU run needs the stand-alone driver binary, not found: " + exe);
        // rendezvous file of the communicator id: a name nobody can predict (mkstemp), removed again so that the ranks see it appear
        char id_tmpl[] = "/tmp/rrx_comm_XXXXXX";
        const int id_fd = mkstemp(id_tmpl);
        if (id_fd < 0) throw std::runtime_error("cannot create the rendezvous file of the communicator id");
        close(id_fd);
        const std::string id_file = std::string(id_tmpl) + ".id";
        std::remove(id_tmpl);
        std::remove(id_file.c_str());
        std::vector<pid_t> pids;
        for (int r=0; r<n; ++r)
        {
            std::vector<std::string> env_s;
            for (char** e = environ; *e; ++e) env_s.emplace_back(*e);
            env_s.push_back("RRX_RANK=" + std::to_string(r));
            env_s.push_back("RRX_WORLD=" + std::to_string(n));
            env_s.push_back("RRX_COMM_FILE=" + id_file);
            if (!std::getenv("HSA_ENABLE_IPC_MODE_LEGACY")) env_s.push_back("HSA_ENABLE_IPC_MODE_LEGACY=0");
            std::vector<char*> envp, argvp;
            for (auto& e : env_s) envp.push_back(const_cast<char*>(e.c_str()));
            envp.push_back(nullptr);
            std::vector<std::string> a = args;
            a.insert(a.begin(), exe);
            for (auto& x : a) argvp.push_back(const_cast<char*>(x.c_str()));
            argvp.push_back(nullptr);
            pid_t pid;
            if (posix_spawn(&pid, exe.c_str(), nullptr, nullptr, argvp.data(), envp.data()) != 0)
            {
                for (const pid_t p : pids) kill(p, SIGTERM);
                for (const pid_t p : pids) waitpid(p, nullptr, 0);
                throw std::runtime_error("cannot start rank " + std::to_string(r));
            }
            pids.push_back(pid);
        }
        // Reap in completion order. A rank that fails (non-zero exit, signal) leaves the others blocked for ever in
        // ncclCommInitRank / ncclAllGather, which have no timeout: the survivors are terminated (SIGTERM, then SIGKILL after a
        // grace period) and the launcher reports the failure instead of hanging.
        int worst = 0;
        size_t left = pids.size();
        bool failed = false;
        while (left > 0)
        {
            int st = 0;
            const pid_t done = waitpid(-1, &st, 0);
            if (done < 0) { if (errno == EINTR) continue; break; }
            const auto it = std::find(pids.begin(), pids.end(), done);
            if (it == pids.end()) continue;                       // not one of ours
            *it = -1; --left;
            const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
            worst = std::max(worst, rc);
            if (rc != 0 && !failed)
            {
                failed = true;
                Status::print_error("rank process " + std::to_string(int(done)) + " failed (status " + std::to_string(rc) + "): stopping the other ranks");
                for (const pid_t p : pids) if (p > 0) kill(p, SIGTERM);
                for (int tick=0; tick<50 && left > 0; ++tick)     // up to 5 s for a clean exit
                {
                    for (pid_t& p : pids)
                        if (p > 0 && waitpid(p, &st, WNOHANG) == p) { p = -1; --left; }
                    if (left > 0) usleep(100000);
                }
                for (const pid_t p : pids) if (p > 0) kill(p, SIGKILL);
            }
        }
        std::remove(id_file.c_str());
        return worst;
    }

    // ---- column order (--sort-columns). Columns are independent, so the order in which they are solved is free; the windowed gas
    // optics wants neighbouring columns to be alike (with surface pressures 35 % apart every workgroup falls back to
    // the gather kernels, 1.4 x the time). The driver therefore solves the columns in ascending order of surface pressure where
    // they differ by more than a cell of the k-distribution's pressure grid inside a 256-column stretch, and writes every output
    // in the order of the input file. `perm[i]` = input column (0-based) solved at position i; empty = input order.
    std::vector<int> column_order(const Array<Float,2>& p_lev, const bool on)
    {
        const int n_col = p_lev.dim(1), n_lev = p_lev.dim(2);
        std::vector<Float> p_sfc(n_col);
        for (int i=0; i<n_col; ++i) p_sfc[i] = std::max(p_lev.v()[i], p_lev.v()[i + size_t(n_lev-1)*n_col]);
        bool spread = false;
        for (int b=0; b<n_col && !spread; b+=256)
        {
            const auto mm = std::minmax_element(p_sfc.begin() + b, p_sfc.begin() + std::min(n_col, b + 256));
            spread = *mm.second > Float(1.2) * *mm.first;
        }
        if (!on || !spread) return {};
        std::vector<int> perm(n_col);
        std::iota(perm.begin(), perm.end(), 0);
        std::stable_sort(perm.begin(), perm.end(), [&](const int a, const int b) { return p_sfc[a] < p_sfc[b]; });
        return perm;
    }
    // input arrays into solve order: columns in the first dimension (n_col, ...) or in the last of two (n_bnd, n_col)
    template<int N> Array<Float,N> sorted_first(const Array<Float,N>& a, const std::vector<int>& perm)
    {
        if (perm.empty() || a.size() == 0 || a.dim(1) != int(perm.size())) return a;
        const size_t n_col = perm.size(), m = size_t(a.size()) / n_col;
        Array<Float,N> out(a.get_dims());
        for (size_t k=0; k<m; ++k)
            for (size_t i=0; i<n_col; ++i) out.v()[i + k*n_col] = a.v()[perm[i] + k*n_col];
        return out;
    }
    Array<Float,2> sorted_last(const Array<Float,2>& a, const std::vector<int>& perm)
    {
        if (perm.empty() || a.size() == 0 || a.dim(2) != int(perm.size())) return a;
        const size_t n1 = a.dim(1);
        Array<Float,2> out(a.get_dims());
        for (size_t i=0; i<perm.size(); ++i)
            for (size_t k=0; k<n1; ++k) out.v()[k + i*n1] = a.v()[k + size_t(perm[i])*n1];
        return out;
    }
    // output arrays (columns first, all columns) back into the order of the input file
    template<int N> Array<Float,N> input_order(Array<Float,N> a, const std::vector<int>& perm)
    {
        if (perm.empty() || a.size() == 0 || a.dim(1) != int(perm.size())) return a;
        const size_t n_col = perm.size(), m = size_t(a.size()) / n_col;
        Array<Float,N> out(a.get_dims());
        for (size_t k=0; k<m; ++k)
            for (size_t i=0; i<n_col; ++i) out.v()[perm[i] + k*n_col] = a.v()[i + k*n_col];
        return out;
    }

    void read_and_set_vmr(const std::string& gas_name, const int n_col_x, const int n_col_y, const int n_lay,
                          const Netcdf_handle& input_nc, Gas_concs& gas_concs)
    {
        const std::string vmr_gas_name = "vmr_" + gas_name;
        if (!input_nc.variable_exists(vmr_gas_name))
        {
            Status::print_warning("Gas \"" + gas_name + "\" not available in input file.");
            return;
        }
        const std::map<std::string, int> dims = input_nc.get_variable_dimensions(vmr_gas_name);
        const int n_dims = int(dims.size());
        if (n_dims == 0)
            gas_concs.set_vmr(gas_name, input_nc.get_variable<Float>(vmr_gas_name));
        else if (n_dims == 1)
        {
            if (dims.count("lay") == 0) throw std::runtime_error("Illegal dimensions of gas \"" + gas_name + "\" in input");
            gas_concs.set_vmr(gas_name, Array<Float,1>(input_nc.get_variable<Float>(vmr_gas_name, {n_lay}), {n_lay}));
        }
        else if (n_dims == 3)
            gas_concs.set_vmr(gas_name, Array<Float,2>(input_nc.get_variable<Float>(vmr_gas_name, {n_lay, n_col_y, n_col_x}), {n_col_x*n_col_y, n_lay}));
        else
            throw std::runtime_error("Illegal dimensions of gas \"" + gas_name + "\" in input");
    }

    bool parse_command_line_options(std::map<std::string, std::pair<bool, std::string>>& options, int argc, char** argv)
    {
        for (int i=1; i<argc; ++i)
        {
            std::string argument(argv[i]);
            if (argument == "-h" || argument == "--help")
            {
                Status::print_message("Possible usage:");
                for (const auto& clo : options)
                {
                    std::ostringstream ss;
                    ss << std::left << std::setw(30) << ("--" + clo.first) << clo.second.second;
                    Status::print_message(ss.str());
                }
                return true;
            }
            if (argument.size() < 3 || argument[0] != '-' || argument[1] != '-')
                throw std::runtime_error(argument + " is an illegal command line option.");
            argument.erase(0, 2);
            bool enable = true;
            if (argument.compare(0, 3, "no-") == 0) { enable = false; argument.erase(0, 3); }
            if (options.find(argument) == options.end())
                throw std::runtime_error(argument + " is an illegal command line option.");
            options.at(argument).first = enable;
        }
        return false;
    }

    double now_ms()
    {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
}


void solve_radiation(int argc, char** argv)
{
    Status::print_message("###### Starting RTE+RRTMGP solver ######");

    std::map<std::string, std::pair<bool, std::string>> command_line_options {
        {"shortwave"        , { true,  "Enable computation of shortwave radiation."}},
        {"longwave"         , { true,  "Enable computation of longwave radiation." }},
        {"fluxes"           , { true,  "Enable computation of fluxes."             }},
        {"cloud-optics"     , { false, "Enable cloud optics."                      }},
        {"aerosol-optics"   , { false, "Enable aerosol optics."                       }},
        {"output-optical"   , { false, "Enable output of optical properties."      }},
        {"output-bnd-fluxes", { false, "Enable output of band fluxes."             }},
        {"timings"          , { false, "Repeat computation 10x for run times."     }},
        {"delta-cloud"      , { true,  "delta-scaling of cloud optical properties"   }},
        {"delta-aerosol"    , { false, "delta-scaling of aerosol optical properties" }},
        {"broadband-solvers", { true,  "Sum g-points inside the solvers (no per-g-point fluxes; off with --output-bnd-fluxes)." }},
        {"heating-rates"    , { false, "Output layer heating rates lw_heating_rate / sw_heating_rate (K/s)." }},
        {"async"            , { false, "Host-model mode: vertical ordering read once, solves enqueued without synchronising." }},
        {"sort-columns"     , { true,  "Solve the columns in order of surface pressure where neighbours differ much (outputs keep the input order)." }},
        {"device-sort-columns", { false, "Leave the ordering to the solvers: sorted and padded on the device inside solve_gpu (Radiation_solver::set_column_sorting(1)) instead of on the host before the upload." }}};

    if (parse_command_line_options(command_line_options, argc, argv))
        return;

    const bool switch_shortwave         = command_line_options.at("shortwave"        ).first;
    const bool switch_longwave          = command_line_options.at("longwave"         ).first;
    const bool switch_fluxes            = command_line_options.at("fluxes"           ).first;
    const bool switch_cloud_optics      = command_line_options.at("cloud-optics"     ).first;
    const bool switch_aerosol_optics    = command_line_options.at("aerosol-optics"   ).first;
    const bool switch_output_optical    = command_line_options.at("output-optical"   ).first;
    const bool switch_output_bnd_fluxes = command_line_options.at("output-bnd-fluxes").first;
    const bool switch_timings           = command_line_options.at("timings"          ).first;
    const bool switch_delta_cloud       = command_line_options.at("delta-cloud"      ).first;
    const bool switch_delta_aerosol     = command_line_options.at("delta-aerosol"    ).first;
    const bool switch_broadband         = command_line_options.at("broadband-solvers").first;
    const bool switch_heating_rates     = command_line_options.at("heating-rates"    ).first;
    const bool switch_async             = command_line_options.at("async"            ).first;
    const bool switch_device_sort       = command_line_options.at("device-sort-columns").first;
    const bool switch_sort_columns      = command_line_options.at("sort-columns"     ).first && !switch_device_sort;

    Status::print_message("Solver settings:");
    for (const auto& option : command_line_options)
    {
        std::ostringstream ss;
        ss << std::left << std::setw(20) << option.first << " = " << std::boolalpha << option.second.first;
        Status::print_message(ss.str());
    }
    const int col_block = std::getenv("RRX_COL_BLOCK") ? std::atoi(std::getenv("RRX_COL_BLOCK")) : 16384;

    ////// READ THE ATMOSPHERIC DATA //////
    Status::print_message("Reading atmospheric input data from NetCDF.");
    Netcdf_file input_nc("rte_rrtmgp_input.nc", Netcdf_mode::Read);
    const int n_col_x = input_nc.get_dimension_size("x");
    const int n_col_y = input_nc.get_dimension_size("y");
    const int n_col_glob = n_col_x * n_col_y;

    // one rank per GPU (RRX_RANK / RRX_WORLD set by the --ngpus launcher): this rank's contiguous column range
    Ranks ranks;
    ranks.init(std::getenv("RRX_WORLD") ? std::atoi(std::getenv("RRX_WORLD")) : 1, std::getenv("RRX_RANK") ? std::atoi(std::getenv("RRX_RANK")) : 0, n_col_glob);
    const int n_col = ranks.col_e - ranks.col_s;
    const bool sharded = ranks.world > 1;
    if (sharded)
        Status::print_message("Rank " + std::to_string(ranks.rank) + " of " + std::to_string(ranks.world) + ": columns " +
                              std::to_string(ranks.col_s + 1) + " - " + std::to_string(ranks.col_e));
    const int n_lay = input_nc.get_dimension_size("lay");
    const int n_lev = input_nc.get_dimension_size("lev");

    const Array<Float,2> p_lay_all(input_nc.get_variable<Float>("p_lay", {n_lay, n_col_y, n_col_x}), {n_col_glob, n_lay});
    const Array<Float,2> p_lev_all(input_nc.get_variable<Float>("p_lev", {n_lev, n_col_y, n_col_x}), {n_col_glob, n_lev});
    // solve order of the columns (every rank computes the same one from the whole file, then takes its range of it)
    const std::vector<int> perm = column_order(p_lev_all, switch_sort_columns);
    if (!perm.empty()) Status::print_message("Columns are solved in order of surface pressure (--no-sort-columns keeps the input order).");
    auto shard2 = [&](const Array<Float,2>& a0)
    {
        const Array<Float,2> a = sorted_first(a0, perm);
        return (!sharded || a.size() == 0) ? a : a.subset({{ {ranks.col_s + 1, ranks.col_e}, {1, a.dim(2)} }});
    };
    auto shard_last = [&](const Array<Float,2>& a0)
    {
        const Array<Float,2> a = sorted_last(a0, perm);
        return !sharded ? a : a.subset({{ {1, a.dim(1)}, {ranks.col_s + 1, ranks.col_e} }});
    };
    auto shard1 = [&](const Array<Float,1>& a0)
    {
        const Array<Float,1> a = sorted_first(a0, perm);
        return !sharded ? a : a.subset({{ {ranks.col_s + 1, ranks.col_e} }});
    };
    Array<Float,2> p_lay = shard2(p_lay_all);
    Array<Float,2> t_lay = shard2(Array<Float,2>(input_nc.get_variable<Float>("t_lay", {n_lay, n_col_y, n_col_x}), {n_col_glob, n_lay}));
    Array<Float,2> p_lev = shard2(p_lev_all);
    Array<Float,2> t_lev = shard2(Array<Float,2>(input_nc.get_variable<Float>("t_lev", {n_lev, n_col_y, n_col_x}), {n_col_glob, n_lev}));

    Array<Float,2> col_dry;
    if (input_nc.variable_exists("col_dry"))
        col_dry = shard2(Array<Float,2>(input_nc.get_variable<Float>("col_dry", {n_lay, n_col_y, n_col_x}), {n_col_glob, n_lay}));

    Gas_concs gas_concs_all;
    for (const char* gas : {"h2o", "co2", "o3", "n2o", "co", "ch4", "o2", "n2", "ccl4", "cfc11", "cfc12", "cfc22",
                            "hfc143a", "hfc125", "hfc23", "hfc32", "hfc134a", "cf4", "no2"})
        read_and_set_vmr(gas, n_col_x, n_col_y, n_lay, input_nc, gas_concs_all);
    if (!perm.empty())
        for (const char* gas : {"h2o", "co2", "o3", "n2o", "co", "ch4", "o2", "n2", "ccl4", "cfc11", "cfc12", "cfc22",
                                "hfc143a", "hfc125", "hfc23", "hfc32", "hfc134a", "cf4", "no2"})
            if (gas_concs_all.exists(gas) && gas_concs_all.get_vmr(gas).dim(1) == n_col_glob)
                gas_concs_all.set_vmr(gas, sorted_first(Array<Float,2>(gas_concs_all.get_vmr(gas)), perm));
    const Gas_concs gas_concs = sharded ? Gas_concs(gas_concs_all, ranks.col_s + 1, n_col) : gas_concs_all;

    Array<Float,2> lwp, iwp, rel, dei;
    if (switch_cloud_optics)
    {
        lwp = shard2(Array<Float,2>(input_nc.get_variable<Float>("lwp", {n_lay, n_col_y, n_col_x}), {n_col_glob, n_lay}));
        iwp = shard2(Array<Float,2>(input_nc.get_variable<Float>("iwp", {n_lay, n_col_y, n_col_x}), {n_col_glob, n_lay}));
        rel = shard2(Array<Float,2>(input_nc.get_variable<Float>("rel", {n_lay, n_col_y, n_col_x}), {n_col_glob, n_lay}));
        dei = shard2(Array<Float,2>(input_nc.get_variable<Float>("dei", {n_lay, n_col_y, n_col_x}), {n_col_glob, n_lay}));
    }

    ////// CREATE THE OUTPUT FILE //////
    Status::print_message("Preparing NetCDF output file.");
    // every rank runs the same output code on the gathered arrays; only rank 0's file is kept
    const std::string output_name = ranks.rank == 0 ? "rte_rrtmgp_output.nc" : "rte_rrtmgp_output.nc.rank" + std::to_string(ranks.rank);
    Netcdf_file output_nc(output_name, Netcdf_mode::Create);
    output_nc.add_dimension("x", n_col_x);
    output_nc.add_dimension("y", n_col_y);
    output_nc.add_dimension("lay", n_lay);
    output_nc.add_dimension("lev", n_lev);
    output_nc.add_dimension("pair", 2);
    output_nc.add_variable<Float>("p_lay", {"lay", "y", "x"}).insert(p_lay_all.v(), {0, 0, 0});
    output_nc.add_variable<Float>("p_lev", {"lev", "y", "x"}).insert(p_lev_all.v(), {0, 0, 0});

    Gas_concs_gpu gas_concs_gpu(gas_concs);
    Array_gpu<Float,2> p_lay_gpu(p_lay), p_lev_gpu(p_lev), t_lay_gpu(t_lay), t_lev_gpu(t_lev), col_dry_gpu(col_dry);
    Array_gpu<Float,2> lwp_gpu(lwp), iwp_gpu(iwp), rel_gpu(rel), dei_gpu(dei);

    auto time_runs = [&](const std::string& name, const std::function<void()>& run)
    {
        run();                                               // warm-up (allocations, first launches)
        rrx_host::check(rrx_synchronize(nullptr));
        const int n_runs = switch_timings ? 11 : 1;
        for (int i=0; i<n_runs; ++i)
        {
            const double t0 = now_ms();
            run();
            rrx_host::check(rrx_synchronize(nullptr));
            Status::print_message("Duration " + name + " solver: " + std::to_string(now_ms() - t0) + " (ms)");
        }
    };

    ////// RUN THE LONGWAVE SOLVER //////
    if (switch_longwave)
    {
        Status::print_message("Initializing the longwave solver.");
        Radiation_solver_longwave rad_lw(gas_concs_gpu, "coefficients_lw.nc", switch_cloud_optics ? "cloud_coefficients_lw.nc" : "");
        rad_lw.set_column_block(col_block);
        rad_lw.set_broadband_solvers(switch_broadband);
        // (--no-sort-columns: the file's order and column count exactly; otherwise the solver pads to a multiple of 16 columns and,
        //  with --device-sort-columns, orders them itself)
        rad_lw.set_column_sorting(switch_device_sort ? 1 : (switch_sort_columns ? -1 : 0));
        rad_lw.set_column_padding(switch_sort_columns || switch_device_sort);
        if (switch_async) rad_lw.set_vertical_ordering(p_lay({1, 1}) < p_lay({1, n_lay}) ? 1 : 0);    // known on the host: no read-backs per solve

        const int n_bnd_lw = rad_lw.get_n_bnd_gpu();
        const int n_gpt_lw = rad_lw.get_n_gpt_gpu();
        Array<Float,2> emis_sfc = shard_last(Array<Float,2>(input_nc.get_variable<Float>("emis_sfc", {n_col_y, n_col_x, n_bnd_lw}), {n_bnd_lw, n_col_glob}));
        Array<Float,1> t_sfc = shard1(Array<Float,1>(input_nc.get_variable<Float>("t_sfc", {n_col_y, n_col_x}), {n_col_glob}));
        Array_gpu<Float,2> emis_sfc_gpu(emis_sfc);
        Array_gpu<Float,1> t_sfc_gpu(t_sfc);

        Array_gpu<Float,3> lw_tau, lay_source, lev_source;
        Array_gpu<Float,2> sfc_source;
        if (switch_output_optical)
        {
            lw_tau.set_dims({n_col, n_lay, n_gpt_lw}); lay_source.set_dims({n_col, n_lay, n_gpt_lw});
            lev_source.set_dims({n_col, n_lev, n_gpt_lw}); sfc_source.set_dims({n_col, n_gpt_lw});
        }
        Array_gpu<Float,2> lw_flux_up, lw_flux_dn, lw_flux_net;
        if (switch_fluxes) { lw_flux_up.set_dims({n_col, n_lev}); lw_flux_dn.set_dims({n_col, n_lev}); lw_flux_net.set_dims({n_col, n_lev}); }
        Array_gpu<Float,3> lw_bnd_flux_up, lw_bnd_flux_dn, lw_bnd_flux_net;
        if (switch_output_bnd_fluxes)
        {
            lw_bnd_flux_up.set_dims({n_col, n_lev, n_bnd_lw}); lw_bnd_flux_dn.set_dims({n_col, n_lev, n_bnd_lw}); lw_bnd_flux_net.set_dims({n_col, n_lev, n_bnd_lw});
        }

        Status::print_message("Solving the longwave radiation.");
        time_runs("longwave", [&]()
        {
            rad_lw.solve_gpu(switch_fluxes, switch_cloud_optics, switch_output_optical, switch_output_bnd_fluxes,
                    gas_concs_gpu, p_lay_gpu, p_lev_gpu, t_lay_gpu, t_lev_gpu, col_dry_gpu, t_sfc_gpu, emis_sfc_gpu,
                    lwp_gpu, iwp_gpu, rel_gpu, dei_gpu, lw_tau, lay_source, lev_source, sfc_source,
                    lw_flux_up, lw_flux_dn, lw_flux_net, lw_bnd_flux_up, lw_bnd_flux_dn, lw_bnd_flux_net);
        });

        Status::print_message("Storing the longwave output.");
        output_nc.add_dimension("gpt_lw", n_gpt_lw);
        output_nc.add_dimension("band_lw", n_bnd_lw);
        output_nc.add_variable<Float>("lw_band_lims_wvn", {"band_lw", "pair"}).insert(rad_lw.get_band_lims_wavenumber_gpu().v(), {0, 0});
        if (switch_output_optical)
        {
            output_nc.add_variable<int>("lw_band_lims_gpt", {"band_lw", "pair"}).insert(rad_lw.get_band_lims_gpoint_gpu().v(), {0, 0});
            output_nc.add_variable<Float>("lw_tau", {"gpt_lw", "lay", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(lw_tau, n_col_glob)), perm).v(), {0, 0, 0, 0});
            output_nc.add_variable<Float>("lay_source", {"gpt_lw", "lay", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(lay_source, n_col_glob)), perm).v(), {0, 0, 0, 0});
            output_nc.add_variable<Float>("lev_source", {"gpt_lw", "lev", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(lev_source, n_col_glob)), perm).v(), {0, 0, 0, 0});
            output_nc.add_variable<Float>("sfc_source", {"gpt_lw", "y", "x"}).insert(input_order(Array<Float,2>(ranks.gather(sfc_source, n_col_glob)), perm).v(), {0, 0, 0});
        }
        if (switch_fluxes)
        {
            output_nc.add_variable<Float>("lw_flux_up" , {"lev", "y", "x"}).insert(input_order(Array<Float,2>(ranks.gather(lw_flux_up, n_col_glob)), perm).v(), {0, 0, 0});
            output_nc.add_variable<Float>("lw_flux_dn" , {"lev", "y", "x"}).insert(input_order(Array<Float,2>(ranks.gather(lw_flux_dn, n_col_glob)), perm).v(), {0, 0, 0});
            output_nc.add_variable<Float>("lw_flux_net", {"lev", "y", "x"}).insert(input_order(Array<Float,2>(ranks.gather(lw_flux_net, n_col_glob)), perm).v(), {0, 0, 0});
            if (switch_heating_rates)
            {
                Array_gpu<Float,2> hr;
                compute_heating_rate(lw_flux_net, p_lev_gpu, hr);
                output_nc.add_variable<Float>("lw_heating_rate", {"lay", "y", "x"}).insert(input_order(Array<Float,2>(ranks.gather(hr, n_col_glob)), perm).v(), {0, 0, 0});
            }
            if (switch_output_bnd_fluxes)
            {
                output_nc.add_variable<Float>("lw_bnd_flux_up" , {"band_lw", "lev", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(lw_bnd_flux_up, n_col_glob)), perm).v(), {0, 0, 0, 0});
                output_nc.add_variable<Float>("lw_bnd_flux_dn" , {"band_lw", "lev", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(lw_bnd_flux_dn, n_col_glob)), perm).v(), {0, 0, 0, 0});
                output_nc.add_variable<Float>("lw_bnd_flux_net", {"band_lw", "lev", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(lw_bnd_flux_net, n_col_glob)), perm).v(), {0, 0, 0, 0});
            }
        }
    }

    ////// RUN THE SHORTWAVE SOLVER //////
    if (switch_shortwave)
    {
        Status::print_message("Initializing the shortwave solver.");
        Radiation_solver_shortwave rad_sw(gas_concs_gpu, switch_cloud_optics, switch_aerosol_optics,
                "coefficients_sw.nc", "cloud_coefficients_sw.nc", "aerosol_optics.nc");
        rad_sw.set_column_block(col_block);
        rad_sw.set_broadband_solvers(switch_broadband);
        rad_sw.set_column_sorting(switch_device_sort ? 1 : (switch_sort_columns ? -1 : 0));
        rad_sw.set_column_padding(switch_sort_columns || switch_device_sort);
        if (switch_async) rad_sw.set_vertical_ordering(p_lay({1, 1}) < p_lay({1, n_lay}) ? 1 : 0);

        const int n_bnd_sw = rad_sw.get_n_bnd_gpu();
        const int n_gpt_sw = rad_sw.get_n_gpt_gpu();
        Array<Float,1> mu0 = shard1(Array<Float,1>(input_nc.get_variable<Float>("mu0", {n_col_y, n_col_x}), {n_col_glob}));
        Array<Float,2> sfc_alb_dir = shard_last(Array<Float,2>(input_nc.get_variable<Float>("sfc_alb_dir", {n_col_y, n_col_x, n_bnd_sw}), {n_bnd_sw, n_col_glob}));
        Array<Float,2> sfc_alb_dif = shard_last(Array<Float,2>(input_nc.get_variable<Float>("sfc_alb_dif", {n_col_y, n_col_x, n_bnd_sw}), {n_bnd_sw, n_col_glob}));

        Array<Float,1> tsi_scaling({n_col});
        if (input_nc.variable_exists("tsi"))
        {
            Array<Float,1> tsi = shard1(Array<Float,1>(input_nc.get_variable<Float>("tsi", {n_col_y, n_col_x}), {n_col_glob}));
            const Float tsi_ref = rad_sw.get_tsi_gpu();
            for (int icol=1; icol<=n_col; ++icol) tsi_scaling({icol}) = tsi({icol}) / tsi_ref;
        }
        else if (input_nc.variable_exists("tsi_scaling"))
        {
            const Float tsi_scaling_in = input_nc.get_variable<Float>("tsi_scaling");
            for (int icol=1; icol<=n_col; ++icol) tsi_scaling({icol}) = tsi_scaling_in;
        }
        else
            for (int icol=1; icol<=n_col; ++icol) tsi_scaling({icol}) = Float(1.);

        Array_gpu<Float,1> mu0_gpu(mu0), tsi_scaling_gpu(tsi_scaling);
        Array_gpu<Float,2> sfc_alb_dir_gpu(sfc_alb_dir), sfc_alb_dif_gpu(sfc_alb_dif), rh_gpu;
        // /root/reference/src_test/test_rte_rrtmgp.cu:72-103,303-320: relative humidity and the 11 CAMS mixing ratios, each
        // either an (n_lay) profile or an (n_lay, y, x) field
        Aerosol_concs aerosol_concs;
        if (switch_aerosol_optics)
        {
            rh_gpu = shard2(Array<Float,2>(input_nc.get_variable<Float>("rh", {n_lay, n_col_y, n_col_x}), {n_col_glob, n_lay}));
            for (int i=1; i<=11; ++i)
            {
                const std::string name = std::string(i < 10 ? "aermr0" : "aermr") + std::to_string(i);
                if (!input_nc.variable_exists(name))
                    throw std::runtime_error("Aerosol type \"" + name + "\" not available in input file.");
                const std::map<std::string, int> dims = input_nc.get_variable_dimensions(name);
                if (dims.size() == 1 && dims.count("lay"))
                    aerosol_concs.set_vmr(name, Array<Float,1>(input_nc.get_variable<Float>(name, {n_lay}), {n_lay}));
                else if (dims.size() == 3 && dims.count("lay") && dims.count("y") && dims.count("x"))
                    aerosol_concs.set_vmr(name, sorted_first(Array<Float,2>(input_nc.get_variable<Float>(name, {n_lay, n_col_y, n_col_x}), {n_col_glob, n_lay}), perm));
                else
                    throw std::runtime_error("Illegal dimensions of \"" + name + "\" in input");
            }
        }
        Aerosol_concs_gpu aerosol_concs_gpu(sharded ? Aerosol_concs(aerosol_concs, ranks.col_s + 1, n_col) : aerosol_concs);

        Array_gpu<Float,3> sw_tau, ssa, g;
        Array_gpu<Float,2> toa_src;
        if (switch_output_optical)
        {
            sw_tau.set_dims({n_col, n_lay, n_gpt_sw}); ssa.set_dims({n_col, n_lay, n_gpt_sw}); g.set_dims({n_col, n_lay, n_gpt_sw});
            toa_src.set_dims({n_col, n_gpt_sw});
        }
        Array_gpu<Float,2> sw_flux_up, sw_flux_dn, sw_flux_dn_dir, sw_flux_net;
        if (switch_fluxes)
        {
            sw_flux_up.set_dims({n_col, n_lev}); sw_flux_dn.set_dims({n_col, n_lev}); sw_flux_dn_dir.set_dims({n_col, n_lev}); sw_flux_net.set_dims({n_col, n_lev});
        }
        Array_gpu<Float,3> sw_bnd_flux_up, sw_bnd_flux_dn, sw_bnd_flux_dn_dir, sw_bnd_flux_net;
        if (switch_output_bnd_fluxes)
        {
            sw_bnd_flux_up.set_dims({n_col, n_lev, n_bnd_sw}); sw_bnd_flux_dn.set_dims({n_col, n_lev, n_bnd_sw});
            sw_bnd_flux_dn_dir.set_dims({n_col, n_lev, n_bnd_sw}); sw_bnd_flux_net.set_dims({n_col, n_lev, n_bnd_sw});
        }

        Status::print_message("Solving the shortwave radiation.");
        time_runs("shortwave", [&]()
        {
            rad_sw.solve_gpu(switch_fluxes, switch_cloud_optics, switch_aerosol_optics, switch_output_optical, switch_output_bnd_fluxes,
                    switch_delta_cloud, switch_delta_aerosol, gas_concs_gpu, p_lay_gpu, p_lev_gpu, t_lay_gpu, t_lev_gpu, col_dry_gpu,
                    sfc_alb_dir_gpu, sfc_alb_dif_gpu, tsi_scaling_gpu, mu0_gpu, lwp_gpu, iwp_gpu, rel_gpu, dei_gpu, rh_gpu, aerosol_concs_gpu,
                    sw_tau, ssa, g, toa_src, sw_flux_up, sw_flux_dn, sw_flux_dn_dir, sw_flux_net,
                    sw_bnd_flux_up, sw_bnd_flux_dn, sw_bnd_flux_dn_dir, sw_bnd_flux_net);
        });

        Status::print_message("Storing the shortwave output.");
        output_nc.add_dimension("gpt_sw", n_gpt_sw);
        output_nc.add_dimension("band_sw", n_bnd_sw);
        output_nc.add_variable<Float>("sw_band_lims_wvn", {"band_sw", "pair"}).insert(rad_sw.get_band_lims_wavenumber_gpu().v(), {0, 0});
        if (switch_output_optical)
        {
            output_nc.add_variable<int>("sw_band_lims_gpt", {"band_sw", "pair"}).insert(rad_sw.get_band_lims_gpoint_gpu().v(), {0, 0});
            output_nc.add_variable<Float>("sw_tau", {"gpt_sw", "lay", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(sw_tau, n_col_glob)), perm).v(), {0, 0, 0, 0});
            output_nc.add_variable<Float>("ssa", {"gpt_sw", "lay", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(ssa, n_col_glob)), perm).v(), {0, 0, 0, 0});
            output_nc.add_variable<Float>("g", {"gpt_sw", "lay", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(g, n_col_glob)), perm).v(), {0, 0, 0, 0});
            output_nc.add_variable<Float>("toa_source", {"gpt_sw", "y", "x"}).insert(input_order(Array<Float,2>(ranks.gather(toa_src, n_col_glob)), perm).v(), {0, 0, 0});
        }
        if (switch_fluxes)
        {
            output_nc.add_variable<Float>("sw_flux_up"    , {"lev", "y", "x"}).insert(input_order(Array<Float,2>(ranks.gather(sw_flux_up, n_col_glob)), perm).v(), {0, 0, 0});
            output_nc.add_variable<Float>("sw_flux_dn"    , {"lev", "y", "x"}).insert(input_order(Array<Float,2>(ranks.gather(sw_flux_dn, n_col_glob)), perm).v(), {0, 0, 0});
            output_nc.add_variable<Float>("sw_flux_dn_dir", {"lev", "y", "x"}).insert(input_order(Array<Float,2>(ranks.gather(sw_flux_dn_dir, n_col_glob)), perm).v(), {0, 0, 0});
            output_nc.add_variable<Float>("sw_flux_net"   , {"lev", "y", "x"}).insert(input_order(Array<Float,2>(ranks.gather(sw_flux_net, n_col_glob)), perm).v(), {0, 0, 0});
            if (switch_heating_rates)
            {
                Array_gpu<Float,2> hr;
                compute_heating_rate(sw_flux_net, p_lev_gpu, hr);
                output_nc.add_variable<Float>("sw_heating_rate", {"lay", "y", "x"}).insert(input_order(Array<Float,2>(ranks.gather(hr, n_col_glob)), perm).v(), {0, 0, 0});
            }
            if (switch_output_bnd_fluxes)
            {
                output_nc.add_variable<Float>("sw_bnd_flux_up"    , {"band_sw", "lev", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(sw_bnd_flux_up, n_col_glob)), perm).v(), {0, 0, 0, 0});
                output_nc.add_variable<Float>("sw_bnd_flux_dn"    , {"band_sw", "lev", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(sw_bnd_flux_dn, n_col_glob)), perm).v(), {0, 0, 0, 0});
                output_nc.add_variable<Float>("sw_bnd_flux_dn_dir", {"band_sw", "lev", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(sw_bnd_flux_dn_dir, n_col_glob)), perm).v(), {0, 0, 0, 0});
                output_nc.add_variable<Float>("sw_bnd_flux_net"   , {"band_sw", "lev", "y", "x"}).insert(input_order(Array<Float,3>(ranks.gather(sw_bnd_flux_net, n_col_glob)), perm).v(), {0, 0, 0, 0});
            }
        }
    }
    output_nc.sync();
    if (ranks.rank != 0) std::remove(output_name.c_str());
    Status::print_message("###### Finished RTE+RRTMGP solver ######");
}


// C entry point for tests (ctypes): same behaviour and exit status as main().
extern "C" int rrx_host_main(int argc, char** argv)
{
    try
    {
        std::vector<std::string> args(argv + 1, argv + argc);
        const int n_gpus = extract_ngpus(args);
        if (n_gpus > 1 && !std::getenv("RRX_RANK"))
            return launch_ranks(n_gpus, args);                       // this process only starts and awaits the ranks
        std::vector<char*> av{argv[0]};
        for (auto& a : args) av.push_back(const_cast<char*>(a.c_str()));
        solve_radiation(int(av.size()), av.data());
    }
    catch (const std::exception& e)
    {
        Status::print_error(std::string("EXCEPTION: ") + e.what());
        return 1;
    }
    catch (...)
    {
        Status::print_error("UNHANDLED EXCEPTION!");
        return 1;
    }
    return 0;
}

#ifndef RRX_NO_MAIN
int main(int argc, char** argv) { return rrx_host_main(argc, argv); }
#endif
