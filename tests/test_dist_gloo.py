"""Multi-process CPU test (gloo, world_size 2 and 3) of the N>1 path: column partition + the single all-gather of the
packed broadband fluxes must reproduce the single-process solve bit for bit. The per-rank solve runs on the CPU oracle
here (no GPU in this container); on GPUs the same sharding code runs with the HIP backend and backend "nccl" (bench.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ncol, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_py
        from rte_rrtmgp_cpp_amd import synthetic, pipeline, sharding
        be = oracle_py.CpuKernels("oracle", np.float64)
        kw = dict(ngpt=32, nbnd=2, npres=10, nflav=3, nminor_lower=5, nminor_upper=3)
        kl, ks = be.upload_kdist(synthetic.make_kdist("lw", **kw)), be.upload_kdist(synthetic.make_kdist("sw", **kw))
        full = synthetic.make_atmosphere(ncol, 24, nbnd_lw=2, nbnd_sw=2, seed=11)
        mine = sharding.shard_atmosphere(full, rank, world)
        lw = pipeline.solve_lw(be, kl, mine); sw = pipeline.solve_sw(be, ks, mine)
        packed = np.stack([lw["flux_up"], lw["flux_dn"], lw["flux_net"], sw["flux_up"], sw["flux_dn"], sw["flux_dn_dir"], sw["flux_net"]])
        gathered = sharding.gather_fluxes(torch.from_numpy(packed), ncol).numpy()
        if rank == 0:
            lw = pipeline.solve_lw(be, kl, full); sw = pipeline.solve_sw(be, ks, full)
            ref = np.stack([lw["flux_up"], lw["flux_dn"], lw["flux_net"], sw["flux_up"], sw["flux_dn"], sw["flux_dn_dir"], sw["flux_net"]])
            q.put(bool(np.array_equal(gathered, ref)))
    finally:
        dist.destroy_process_group()


def _bench_worker(rank, world, port, scaling, ncol, q):
    """The sharding + gather code bench.py itself runs for N > 1 (bench.local_atmosphere, sharding.FluxGatherer), with the
    per-rank solve on the CPU oracle: the gathered fluxes must equal the single-rank solve of the same job bit for bit."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import argparse
        import bench, oracle_py
        from rte_rrtmgp_cpp_amd import synthetic, pipeline, sharding
        be = oracle_py.CpuKernels("oracle", np.float64)
        kw = dict(ngpt=32, nbnd=2, npres=10, nflav=3, nminor_lower=5, nminor_upper=3)
        kl, ks = be.upload_kdist(synthetic.make_kdist("lw", **kw)), be.upload_kdist(synthetic.make_kdist("sw", **kw))
        args = argparse.Namespace(ncol=ncol, nlay=20, scaling=scaling)

        def solve(atm):
            lw = pipeline.solve_lw(be, kl, atm, do_broadband=True); sw = pipeline.solve_sw(be, ks, atm, do_broadband=True)
            return np.stack([lw["flux_up"], lw["flux_dn"], lw["flux_net"], sw["flux_up"], sw["flux_dn"], sw["flux_dn_dir"], sw["flux_net"]])

        ntot = bench.global_columns(args, world)
        (s, e), mine = bench.local_atmosphere(args, 2, rank, world)
        assert (s, e) == sharding.column_range(rank, world, ntot) and mine.ncol == e - s
        local = torch.from_numpy(solve(mine))
        gat = sharding.FluxGatherer(ntot, local)
        gat.gather(local)
        got = gat.result().numpy()
        # pipelined use, as in bench.py: the exchange is started and the "next solve" overwrites the source at once; three steps
        # over the two buffer pairs, each result read only after a later gather was started
        src = local.clone()
        first = gat.gather(src); src.mul_(2.0)
        second = gat.gather(src); src.mul_(2.0)
        gat.gather(src); src.zero_()
        got4 = gat.result().numpy()
        nf, nl = local.shape[:2]
        mine_of = lambda buf: buf.view(world, nf, nl, -1)[rank, :, :, :local.shape[-1]]
        pipelined_ok = bool(np.array_equal(got4, 4.0*got) and torch.equal(mine_of(second), 2.0*local) and torch.equal(mine_of(first), 4.0*local))
        sync = sharding.FluxGatherer(ntot, local, pipelined=False)
        sync.gather(local)
        pipelined_ok = pipelined_ok and bool(np.array_equal(sync.result().numpy(), got))
        if rank == 0:
            one = argparse.Namespace(ncol=ntot, nlay=20, scaling="strong")
            (s1, e1), full = bench.local_atmosphere(one, 2, 0, 1)
            q.put(bool((s1, e1) == (0, ntot) and np.array_equal(got, solve(full)) and pipelined_ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,scaling,ncol", [(2, "weak", 6), (2, "strong", 9), (3, "strong", 10)])
def test_bench_sharding_path_on_gloo(world, scaling, ncol, oracle_built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + 7*world + len(scaling)
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, scaling, ncol, q)) for r in range(world)]
    for p in procs: p.start()
    for p in procs: p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=10) is True


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher must spawn N ranks itself: checked through --dry-run,
    which prints the child command instead of running it (no GPU here)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--dry-run"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = out.stdout.strip().split()
    assert "torch.distributed.run" in cmd and "--nproc-per-node=8" in cmd and "--master-addr" in cmd and "127.0.0.1" in cmd
    assert cmd[-5:] == ["--gpus", "8", "--steps", "3", "--dry-run"] and any(c.endswith("bench.py") for c in cmd)
    # N > 1 defaults to BASELINE C4 as worded: 16 384 columns SHARDED over the GPUs (strong scaling)
    assert out.stdout.splitlines()[0].startswith("# scaling strong: 16384 columns over 8 ranks: [2048, 2048")
    weak = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--scaling", "weak", "--dry-run"],
                          capture_output=True, text=True, env=env, timeout=120)
    assert weak.stdout.splitlines()[0].startswith("# scaling weak: 131072 columns over 8 ranks: [16384,")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True, env=env, timeout=120)
    assert one.returncode == 0 and "torch.distributed.run" not in one.stdout


@pytest.mark.parametrize("world,ncol", [(2, 10), (3, 10)])
def test_column_sharding_and_flux_gather(world, ncol, oracle_built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, ncol, q)) for r in range(world)]
    for p in procs: p.start()
    for p in procs: p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=10) is True


def test_all_sky_atmosphere_is_built_per_rank():
    """bench.py --allsky builds only the rank's own columns: the same columns as the shard of the whole job."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    from rte_rrtmgp_cpp_amd import synthetic, sharding
    full = synthetic.make_atmosphere(50, 20, nbnd_lw=2, nbnd_sw=2, seed=1234, clouds=True)
    for world in (1, 3):
        for rank in range(world):
            args = argparse.Namespace(ncol=50, nlay=20, scaling="strong", allsky=True)
            (s, e), part = bench.local_atmosphere(args, 2, rank, world)
            ref = sharding.shard_atmosphere(full, rank, world)
            assert part.ncol == e - s == ref.ncol
            for k in ("lwp", "iwp", "rel", "dei", "t_lay", "p_lay", "t_sfc"):
                assert np.array_equal(getattr(part, k), getattr(ref, k)), k
            assert all(np.array_equal(part.vmr[n], ref.vmr[n]) for n in full.vmr)
    assert (full.lwp > 0).any() and (full.lwp[:, 2] == 0).all()          # every third column is cloud-free


def test_column_ranges_cover_everything():
    from rte_rrtmgp_cpp_amd import sharding
    for n in (1, 7, 16384, 100):
        for w in (1, 2, 3, 8):
            r = [sharding.column_range(k, w, n) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [e - s for s, e in r]
            assert max(sizes) - min(sizes) <= 1
