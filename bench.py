#!/usr/bin/env python
"""bench.py -- columns/sec of one full clear-sky LW+SW solve (140 layers x 256 g-points) on N MI355X.

A step = gas optics (interpolation, major+minor absorption, Rayleigh) -> Planck sources -> lw_solver_noscat ->
sw_solver_2stream -> broadband flux reduction, on a synthetic RCEMIP atmosphere + synthetic k-distribution with
the real shapes; inputs and LUTs are resident in HBM before the timed region starts.
Columns shard over ranks (one process per GPU; for N > 1 the default is strong scaling -- the --ncol columns of C4 split over the
ranks -- and the weak line, --ncol columns PER GPU, rides along as `other_scaling`); the only collective is the
all-gather of the packed broadband fluxes (7 x nlev x ncol words per rank) of each step, which travels while the next step
computes (sharding.FluxGatherer; the last one is awaited inside the timed region).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus 8                      # starts its own 8 ranks (torch.distributed.run as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus 8 --scaling strong     # BASELINE C4 as stated: 16 384 columns sharded over the 8 GPUs

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (dominant kernel, live HIP-event
timing on the launch stream) and `cpu_baseline` (the oracle, a scalar CPU port, on a bounded column sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic words per (column, g-point) at nlay layers -- SURVEY.md section 8(d), DESIGN.md section 5
def algo_words(nlay, ngpt, nbnd, broadband, g_zero, lite):
    nlev = nlay + 1
    nsw = 2 if g_zero else 3        # clear sky: the all-zero asymmetry array is neither written nor read
    if broadband and lite:
        # "Planck-lite" chain: the gas optics writes tau and the Planck fraction per g-point plus the band Planck functions
        # B_lay, B_lev (nbnd/ngpt of a g-point array each); the solver forms the sources from them on chip
        bands = (nlay + nlev) * nbnd / ngpt
        return dict(
            lw_gas_optics=2*nlay + bands + 2,                # tau, pfrac, B_lay, B_lev, sfc_src(+jac)
            lw_planck=0,
            lw_solver=2*nlay + bands + 1 + 2 + 2*nlev/ngpt,  # tau, pfrac, B, secant, sfc_emis + sfc_src, flux sums once
            lw_reduce=0,
            sw_gas_optics=nsw*nlay + 1,
            sw_solver=nsw*nlay + 3 + 3*nlev/ngpt,
            sw_reduce=0)
    if broadband:       # fused form: the solvers keep the g-point sums on chip and store (ncol, nlev) arrays once
        return dict(
            lw_gas_optics=nlay,
            lw_planck=2*nlay + 1 + 2,
            lw_solver=3*nlay + 1 + 2 + 2*nlev/ngpt,
            lw_reduce=0,
            sw_gas_optics=nsw*nlay + 1,
            sw_solver=nsw*nlay + 3 + 3*nlev/ngpt,
            sw_reduce=0)
    return dict(
        lw_gas_optics=nlay,                              # tau
        lw_planck=2*nlay + 1 + 2,                        # lay_src, lev_src, sfc_src(+jac)
        lw_solver=3*nlay + 1 + 2 + 2*nlev,               # 705 at nlay = 140
        lw_reduce=2*nlev,
        sw_gas_optics=3*nlay + 1,                        # tau, ssa, g, toa
        sw_solver=3*nlay + 3 + 3*nlev,                   # 846 at nlay = 140
        sw_reduce=3*nlev)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# vector-ALU issue roof: 256 CUs x 4 SIMDs, one wave64 VALU instruction (fp64 or fp32, non-packed) per SIMD per 4 cycles at the
# 2.4 GHz peak engine clock (MI355X_MICROARCH.md: 78.6 TFLOP/s fp64 vector = this rate x 64 lanes x 2 flops)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4

STAGE_KERNEL = {"lw_solver": ("lw_noscat_bb_kernel", "lw_noscat_scan_kernel"), "sw_solver": ("sw_2stream_scan_kernel",),
                "lw_planck": ("planck_source_kernel",)}


def _pmc_entry(stage, args):
    """Entry of the committed rocprofv3 counter summary of the SAME workload for the stage's kernel (profiles/pmc_traffic.json,
    written by tools/profile_round.sh + tools/pmc_summary.py from separate --pmc passes), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path) or stage not in STAGE_KERNEL:
        return None
    mode = ("broadband" if args.broadband else "per-gpoint") + ("-allsky" if getattr(args, "allsky", False) else "")
    tag = f"{args.dtype}|{mode}|{args.ncol}x{args.nlay}x{args.ngpt}|"
    best = None
    for k, v in json.load(open(path)).items():
        if any(k.startswith(tag + kern) for kern in STAGE_KERNEL[stage]) and "fetch_bytes" in v and "write_bytes" in v:
            if best is None or v["fetch_bytes"] > best["fetch_bytes"]:
                best = v
    return best


def pmc_traffic(stage, args):
    """HBM-side bytes per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE)."""
    v = _pmc_entry(stage, args)
    return None if v is None else int(v["fetch_bytes"] + v["write_bytes"])


def valu_roofline(stage, args, ms):
    """Second roof of a solver kernel: vector-ALU issue. Wave-level VALU instructions per launch come from the committed
    SQ_INSTS_VALU pass of the same workload (a property of the compiled kernel and the shape); the time is measured live."""
    v = _pmc_entry(stage, args)
    if v is None or "insts_valu" not in v:
        return None
    ginst = v["insts_valu"] / (ms*1e-3) / 1e9
    out = {"bound": "valu", "kernel": stage, "achieved": round(ginst, 1), "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s",
           "frac": round(ginst / VALU_PEAK_GINST, 4), "valu_wave_instructions_per_launch": int(v["insts_valu"])}
    if args.dtype == "f32":
        # the nominal roof is one instruction per SIMD every four cycles; measured (tools/issue_mix_bench.hip) an fp32 instruction
        # issues every 2.5 cycles at the three waves per SIMD the fp32 solvers run at: the fraction of THAT rate is the honest one
        out["peak_measured_fp32_3_waves"] = round(VALU_PEAK_GINST * 4.0 / 2.5, 1)
        out["frac_of_measured"] = round(ginst / (VALU_PEAK_GINST * 4.0 / 2.5), 4)
    return out


def cpu_baseline(args, kd_lw0, kd_sw0, be=None):
    """The oracle (scalar C++ port of the reference CPU path, oracle/rrtmgp_oracle.cpp) on a bounded sample of the
    same workload: `cpu_cols` columns in 12-column blocks like src_test/Radiation_solver.cpp:409. `value` is ONE thread (the
    reference CPU executable is single-threaded); `all_cores_value` spreads the same blocks over one process per host core.
    `parity_max_rel`: the first 12-column block solved by the HIP path as well, max |dflux| / max(|flux|, 1) over the
    broadband fluxes (must stay below 1e-6)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    from rte_rrtmgp_cpp_amd import synthetic, pipeline
    if not oracle_py.have("oracle"):
        oracle_py.build(ref=False)
    orc = oracle_py.CpuKernels("oracle", np.float64)
    kd_lw, kd_sw = orc.upload_kdist(kd_lw0), orc.upload_kdist(kd_sw0)
    ncols = args.cpu_cols
    starts = list(range(0, ncols, 12))
    blocks = [synthetic.make_atmosphere(min(12, ncols - c0), args.nlay, nbnd_lw=kd_lw0.nbnd, nbnd_sw=kd_sw0.nbnd, seed=1234 + c0)
              for c0 in starts]
    first = None
    t0 = time.perf_counter()
    for sub in blocks:
        rl = pipeline.solve_lw(orc, kd_lw, sub, do_broadband=True)
        rs = pipeline.solve_sw(orc, kd_sw, sub, do_broadband=True, fused_gas=False)
        if first is None:
            first = (rl, rs)
    dt = time.perf_counter() - t0
    out = dict(value=ncols/dt, unit="columns/s", cores=1, kind="port",
               sample=f"{ncols} columns x {args.nlay} layers x {kd_lw0.ngpt}+{kd_sw0.ngpt} g-points, LW+SW clear-sky, "
                      f"12-column blocks, fp64, broadband mode, {dt:.1f} s")

    if be is not None and be.np_dtype == np.float64:
        sub = pipeline.upload_atmosphere(be, blocks[0])
        gl = pipeline.solve_lw(be, be.upload_kdist(kd_lw0), sub, do_broadband=True)
        gs = pipeline.solve_sw(be, be.upload_kdist(kd_sw0), sub, do_broadband=True)
        worst = 0.0
        for ref, got in ((first[0], gl), (first[1], gs)):
            for k in ("flux_up", "flux_dn", "flux_net"):
                a, b = np.asarray(ref[k], dtype=np.float64), be.to_numpy(got[k]).astype(np.float64)
                worst = max(worst, float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1.0))))
        out["parity_max_rel"] = worst

    ncore = min(os.cpu_count() or 1, 16, len(starts))
    if ncore > 1:
        import multiprocessing as mp
        import cpu_worker
        jobs = [(args.nlay, kd_lw0.ngpt, kd_lw0.nbnd, starts[w::ncore], ncols) for w in range(ncore)]
        with mp.get_context("spawn").Pool(ncore) as pool:          # spawn: the workers never inherit the GPU context
            times = pool.map(cpu_worker.solve_blocks, jobs)
        out["all_cores"] = ncore
        out["all_cores_value"] = ncols / max(times)
    return out


def fp32_against_fp64(args, nbnd, rank, world, solver32, kd_lw0, kd_sw0, ncheck=256):
    """--dtype f32: the first columns of this rank solved once more in fp64 by the same HIP path, and the largest difference of the
    fp32 broadband fluxes from them in W m-2 -- the unit of the reference's own acceptance gate (5.8e-2 W m-2 against its reference
    fluxes, .github/workflows/continuous-integration.yml:58-62). Outside the timed region."""
    import rte_rrtmgp_cpp_amd as R
    from rte_rrtmgp_cpp_amd import synthetic, pipeline, sharding
    ntot = global_columns(args, world)
    s, e = sharding.column_range(rank, world, ntot)
    n = min(ncheck, e - s)
    atm0 = synthetic.make_atmosphere(ntot, args.nlay, nbnd_lw=nbnd, nbnd_sw=nbnd, seed=1234, col_range=(s, s + n),
                                     top_at_1=getattr(args, "top_at_1", False), clouds=getattr(args, "allsky", False))
    if getattr(args, "col_spread", 0.0) > 0:
        atm0 = spread_columns(atm0, args.col_spread, s, s + n, ntot)
    be64 = R.HipKernels(np.float64, solver32.be.device)
    luts = None
    if args.allsky:
        luts = (be64.upload_lut(synthetic.make_cloud_lut(nbnd, "lw")), be64.upload_lut(synthetic.make_cloud_lut(nbnd, "sw")))
    ref = pipeline.ResidentSolver(be64, be64.upload_kdist(kd_lw0), be64.upload_kdist(kd_sw0), pipeline.upload_atmosphere(be64, atm0),
                                  do_broadband=args.broadband, cloud_luts=luts, sort_columns="0").step()
    got = solver32.fluxes[:, :, :n].double()
    d = (got - ref).abs().amax(dim=(1, 2)).tolist()
    names = ("lw_flux_up", "lw_flux_dn", "lw_flux_net", "sw_flux_up", "sw_flux_dn", "sw_flux_dn_dir", "sw_flux_net")
    return {"columns": n, "unit": "W m-2", "max_abs_difference": {k: round(v, 5) for k, v in zip(names, d)},
            "reference_gate": 5.8e-2}


def launch_command(argv, gpus, port=None):
    """What `bench.py --gpus N` (N > 1) runs when it was not started by torch.distributed.run: the documented launcher
    as a CHILD process, one rank per GPU. Nothing in this process has touched the GPU (or imported torch) at that point."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def global_columns(args, world):
    """Total column count of the job: weak scaling keeps --ncol columns PER GPU, strong scaling shards --ncol over the ranks."""
    return args.ncol * world if args.scaling == "weak" else args.ncol


def spread_columns(atm, spread, col_s, col_e, ntot):
    """--col-spread s: every column's pressures scaled by U(1-s, 1+s) and its temperatures shifted by
    U(-30 s, +30 s) K, the spread of tests/test_gpu_parity.py::test_columns_in_different_regimes...; s = 0.35 puts neighbouring
    columns of a wavefront in different LUT cells and regimes. Column c gets the same values whatever the number of ranks."""
    rng = np.random.default_rng(4321)
    scale = rng.uniform(1. - spread, 1. + spread, ntot)[col_s:col_e]
    dT = rng.uniform(-30.*spread, 30.*spread, ntot)[col_s:col_e]
    for k in ("p_lay", "p_lev"):
        setattr(atm, k, np.ascontiguousarray(getattr(atm, k) * scale[None, :]))
    for k in ("t_lay", "t_lev"):
        setattr(atm, k, np.ascontiguousarray(getattr(atm, k) + dT[None, :]))
    atm.t_sfc = np.ascontiguousarray(atm.t_sfc + dT)
    return atm


def local_atmosphere(args, nbnd, rank, world):
    """This rank's column range [start, stop) of the job's global synthetic atmosphere (sharding.column_range) and its
    host-side slice. Column c of the job is the same column whatever the number of ranks."""
    from rte_rrtmgp_cpp_amd import synthetic, sharding
    ntot = global_columns(args, world)
    s, e = sharding.column_range(rank, world, ntot)
    # only this rank's columns are built (column c is the same column whatever the number of ranks), clouds included
    atm = synthetic.make_atmosphere(ntot, args.nlay, nbnd_lw=nbnd, nbnd_sw=nbnd, seed=1234, col_range=(s, e), top_at_1=getattr(args, "top_at_1", False),
                                    clouds=getattr(args, "allsky", False))
    if getattr(args, "col_spread", 0.0) > 0:
        atm = spread_columns(atm, args.col_spread, s, e, ntot)
    return (s, e), atm


def sources_digest():
    """sha256 over the device sources: profiles/pmc_traffic.json records the digest its counters were collected with, so that
    roofline.traffic / roofline_valu can be marked stale once a kernel has changed since."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rte-rrtmgp-cpp_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def pmc_is_stale():
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get("sources_digest", {}).get("sha256") != sources_digest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ncol", type=int, default=128*128, help="columns per GPU (weak scaling) or in total (strong); C4 = 128 x 128")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong (default for --gpus N > 1): --ncol columns sharded over the GPUs (BASELINE C4: 16 384 columns on 8 GPUs = 2 048 each); "
                         "weak: --ncol columns on every GPU. With N > 1 the other one is measured too and reported as `other_scaling`")
    ap.add_argument("--one-scaling", action="store_true", help="N > 1: skip the second measurement (the other scaling)")
    ap.add_argument("--dry-run", action="store_true", help="print the launch command of a multi-GPU run and exit")
    ap.add_argument("--nlay", type=int, default=140)
    ap.add_argument("--ngpt", type=int, default=256)
    ap.add_argument("--top-at-1", action="store_true", help="columns ordered from the top of the atmosphere down (the default is surface first, as the reference's RCEMIP case)")
    ap.add_argument("--nbnd", type=int, default=0, help="bands of the synthetic k-distributions (default ngpt/16; --ngpt 128 --nbnd 16 = the shape of the reduced sets, 8 g-points per band)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--flux-mode", default="broadband", choices=["broadband", "per-gpoint"],
                    help="broadband: do_broadband solvers, g-point sums kept on chip (the CPU path's semantics, default); "
                         "per-gpoint: per-g-point fluxes stored, then sum_broadband (the reference GPU path's flow)")
    ap.add_argument("--broadband", action="store_true", help="same as --flux-mode broadband")
    ap.add_argument("--per-gpoint", action="store_true", help="same as --flux-mode per-gpoint")
    ap.add_argument("--bb-min-groups", type=int, default=None, help="rrx_set_broadband_min_groups (A/B of the fused broadband form)")
    ap.add_argument("--lw-variant", type=int, default=0)
    ap.add_argument("--sw-variant", type=int, default=0)
    ap.add_argument("--cpu-cols", type=int, default=6000, help="columns of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--sync-gather", action="store_true", help="wait for each all-gather before the next solve starts (A/B of the pipelined exchange)")
    ap.add_argument("--overlap", action="store_true", help="run the independent LW and SW chains on two HIP streams")
    ap.add_argument("--col-spread", type=float, default=0.0,
                    help="column heterogeneity: pressures x U(1-s,1+s), temperatures + U(-30s,30s) K per column (default 0 = SURVEY 8(d)'s workload)")
    ap.add_argument("--sort-columns", default=None, choices=["auto", "0", "1"],
                    help="process the columns in ascending order of surface pressure (auto: when neighbouring columns differ; see pipeline.ResidentSolver)")
    ap.add_argument("--driver", default="python", choices=["python", "cxx"],
                    help="python: pipeline.ResidentSolver (every buffer allocated once, the kernels called through the C ABI); cxx: the C++ host "
                         "classes -- Radiation_solver_longwave / _shortwave::solve_gpu, the reference's class structure -- on the same kernels")
    ap.add_argument("--allsky", action="store_true",
                    help="BASELINE's all-sky flow (C5): cloud optics added by band after the gas optics, delta-scaled in SW; not the headline workload")
    args = ap.parse_args()
    args.broadband = (args.flux_mode == "broadband" or args.broadband) and not args.per_gpoint
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"

    # `python bench.py --gpus N` on its own: become the launcher of N ranks. This happens before torch is imported, so this
    # process never initialises the GPU; the ranks are children (never an exec of a process that holds a GPU context).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = launch_command(sys.argv[1:], args.gpus)
        if args.dry_run:
            ntot = global_columns(args, args.gpus)       # (sharding.column_range's split, without importing torch in the launcher)
            print(f"# scaling {args.scaling}: {ntot} columns over {args.gpus} ranks:",
                  [ntot // args.gpus + (1 if r < ntot % args.gpus else 0) for r in range(args.gpus)])
            print(" ".join(cmd)); return 0
        import subprocess
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        return subprocess.run(cmd, env=env).returncode
    if args.dry_run:
        print(" ".join([sys.executable, os.path.abspath(__file__)] + sys.argv[1:])); return 0

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (dmabuf IPC: RCCL between the ranks of a node needs it on this driver)
    import torch
    import torch.distributed as dist
    import rte_rrtmgp_cpp_amd as R
    from rte_rrtmgp_cpp_amd import synthetic, pipeline, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"

    np_dtype = np.float64 if args.dtype == "f64" else np.float32
    be = R.HipKernels(np_dtype, device)
    be.set_variant(lw=args.lw_variant, sw=args.sw_variant)
    if args.bb_min_groups is not None:
        be.set_broadband_min_groups(args.bb_min_groups)
    nbnd = args.nbnd if args.nbnd else args.ngpt // 16
    kd_lw0 = synthetic.make_kdist("lw", ngpt=args.ngpt, nbnd=nbnd)
    kd_sw0 = synthetic.make_kdist("sw", ngpt=args.ngpt, nbnd=nbnd)
    kd_lw, kd_sw = be.upload_kdist(kd_lw0), be.upload_kdist(kd_sw0)
    cloud_luts = None
    if args.allsky:
        cast = lambda lut: be.upload_lut({k: (v.astype(np_dtype) if isinstance(v, np.ndarray) else v) for k, v in lut.items()})
        cloud_luts = (cast(synthetic.make_cloud_lut(nbnd, "lw")), cast(synthetic.make_cloud_lut(nbnd, "sw")))

    def measure(scaling, primary):
        """One job: rank r owns the contiguous column range sharding.column_range(r, world, ntot) of ONE global atmosphere (only its
        own columns are ever built, clouds included). W warm-up steps, then exactly K timed steps between barrier + synchronize
        pairs; the time is the maximum over the ranks."""
        a = argparse.Namespace(**vars(args)); a.scaling = scaling
        ntot = global_columns(a, world)
        (col_s, col_e), atm0 = local_atmosphere(a, nbnd, rank, world)
        atm = pipeline.upload_atmosphere(be, atm0.astype(np_dtype))
        if args.driver == "cxx":
            from rte_rrtmgp_cpp_amd import cxx_driver
            luts0 = (synthetic.make_cloud_lut(nbnd, "lw"), synthetic.make_cloud_lut(nbnd, "sw")) if args.allsky else None
            sort_mode = {None: -1, "auto": -1, "0": 0, "1": 1}[args.sort_columns]
            solver = cxx_driver.CxxDriver(be, kd_lw0, kd_sw0, atm, luts0, column_block=max(atm.ncol, 16), broadband=args.broadband, sort_mode=sort_mode)
        else:
            solver = pipeline.ResidentSolver(be, kd_lw, kd_sw, atm, do_broadband=args.broadband, overlap=args.overlap, cloud_luts=cloud_luts,
                                             sort_columns=args.sort_columns)
        do_gather = world > 1 and not args.no_gather
        gatherer = sharding.FluxGatherer(ntot, solver.fluxes, pipelined=not args.sync_gather) if do_gather else None

        def one_step():
            F = solver.step()
            if gatherer is not None:
                gatherer.gather(F)          # the one collective of the path (tests/test_dist_gloo.py runs this code on gloo)

        for _ in range(args.warmup):
            one_step()
        if gatherer is not None:
            gatherer.finish()
        # one untimed step with the windowed gas optics' hand-back census switched on (it synchronises the stream per launch)
        handed = None
        if primary and rank == 0 and be.lib.has("rrx_gas_window_stats") and args.driver == "python":
            import ctypes
            os.environ["RRX_GW_STATS"] = "1"
            be.lib.cdll.rrx_gas_window_stats(None, None, 1)
            devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2); os.dup2(devnull, 2)      # (the census also prints to stderr)
            try:
                solver.step(); torch.cuda.synchronize()
            finally:
                os.dup2(saved, 2); os.close(saved); os.close(devnull); del os.environ["RRX_GW_STATS"]
            c1, c2 = ctypes.c_longlong(0), ctypes.c_longlong(0)
            be.lib.cdll.rrx_gas_window_stats(ctypes.byref(c1), ctypes.byref(c2), 1)
            if c2.value > 0:
                handed = {"handed_back": int(c1.value), "workgroups": int(c2.value), "frac": round(c1.value / c2.value, 4)}
        if primary and args.driver == "python":
            solver.enable_stage_events(args.steps)

        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        if gatherer is not None:
            gatherer.finish()               # the last exchange is inside the timed region
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dict(dt=dt, ntot=ntot, ncol_local=col_e - col_s, solver=solver, handed=handed)

    m = measure(args.scaling, True)
    dt, ntot, ncol_local, solver, handed = m["dt"], m["ntot"], m["ncol_local"], m["solver"], m["handed"]
    # N > 1: the other way of growing the job, measured in the same run (BASELINE C4 shards 16 384 columns over the
    # GPUs = strong, which is `value`; the weak line -- 16 384 columns on every GPU -- rides along)
    other = None
    if world > 1 and not args.one_scaling:
        o = measure("weak" if args.scaling == "strong" else "strong", False)
        other = {"scaling": "weak" if args.scaling == "strong" else "strong", "value": round(o["ntot"] * args.steps / o["dt"], 1),
                 "ms_per_step": round(o["dt"] / args.steps * 1e3, 3), "columns_per_gpu": o["ncol_local"], "columns_total": o["ntot"]}
        del o

    if rank == 0 and args.driver == "cxx":
        # the C++ host classes enqueue a whole solve per call: no per-stage events from here; the kernels are the ones of the default
        # driver, whose line carries the roofline
        out = {"metric": "columns/sec (LW+SW full solve, 140 lay x 256 gpt)", "value": round(ntot * args.steps / dt, 1), "unit": "columns/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"C4 synthetic {ncol_local} columns/GPU ({ntot} in total) x {args.nlay} layers x {args.ngpt} g-points, "
                                      f"LW+SW {'all-sky' if args.allsky else 'clear-sky'}, RCEMIP profile, synthetic k-distribution (real shapes)",
                          "driver": "C++ host classes (Radiation_solver_longwave / _shortwave::solve_gpu) through librte_rrtmgp_hip.so",
                          "columns_per_gpu": ncol_local, "columns_total": ntot, "nlay": args.nlay, "ngpt": args.ngpt},
               "roofline": None, "finite": bool(torch.isfinite(solver.fluxes).all().item())}
        if other is not None:
            out["other_scaling"] = other
        print(json.dumps(out), flush=True)
    elif rank == 0:
        S = np_dtype().itemsize
        ms = solver.stage_ms()
        words = algo_words(args.nlay, args.ngpt, nbnd, args.broadband, solver.g_zero and args.broadband, solver.lite)
        units = ncol_local * args.ngpt
        kernels = {}
        for st, w in words.items():
            gbs = w * units * S / (max(ms[st], 1e-6)*1e-3) / 1e9
            kernels[st] = dict(ms=round(ms[st], 4), algo_GB=round(w*units*S/1e9, 4), GBs=round(gbs, 1), frac=round(gbs/HBM_PEAK_GBS, 4))
        single = {k: v for k, v in kernels.items() if k in STAGE_KERNEL and v["ms"] > 0}   # single-launch stages
        dom = max(single, key=lambda k: single[k]["ms"])
        finite = bool(torch.isfinite(solver.fluxes).all().item())
        out = {
            "metric": "columns/sec (LW+SW full solve, 140 lay x 256 gpt)",
            "value": round(ntot * args.steps / dt, 1),
            "unit": "columns/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"C4 synthetic {ncol_local} columns/GPU ({ntot} in total) x {args.nlay} layers x {args.ngpt} g-points, "
                                   f"LW+SW {'all-sky (cloud optics, delta-scaled in SW)' if args.allsky else 'clear-sky'}, RCEMIP profile, synthetic k-distribution (real shapes)",
                       "columns_per_gpu": ncol_local, "columns_total": ntot, "nlay": args.nlay, "ngpt": args.ngpt,
                       "flux_mode": "broadband (do_broadband solvers, g-point sums on chip)" if args.broadband
                                    else "per-g-point fluxes + sum_broadband",
                       "col_spread": args.col_spread, "columns_sorted": bool(solver.sort_columns),
                       "parallelism": f"columns sharded x{world}, all-gather of broadband fluxes"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": kernels[dom]["frac"], "traffic": pmc_traffic(dom, args),
                         "algorithmic_bytes_per_launch": int(words[dom]*units*S), "avg_launch_ms": kernels[dom]["ms"]},
            "roofline_valu": valu_roofline(dom, args, kernels[dom]["ms"]),
            "stages": kernels,
            "finite": finite,
        }
        # the PMC-derived entries come from the committed counter summary, not from this run: say so when a kernel has changed since
        stale = pmc_is_stale()
        if stale is not None:
            out["roofline"]["traffic_stale"] = bool(stale)
            if out["roofline_valu"] is not None:
                out["roofline_valu"]["stale"] = bool(stale)
        if handed is not None:
            out["gas_window"] = handed
        if args.dtype == "f32" and args.driver == "python":
            out["fp32_vs_fp64"] = fp32_against_fp64(args, nbnd, rank, world, solver, kd_lw0, kd_sw0)
        if other is not None:
            out["other_scaling"] = other
        if world > 1:       # what a SCALE record can be checked against
            out["ranks_seen"] = dist.get_world_size()
            out["columns_per_rank"] = [int(e - s) for s, e in (sharding.column_range(r, world, ntot) for r in range(world))]
        if world == 1 and args.cpu_cols > 0 and not args.allsky:      # (the CPU sample is the clear-sky headline workload)
            out["cpu_baseline"] = cpu_baseline(args, kd_lw0, kd_sw0, be)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
