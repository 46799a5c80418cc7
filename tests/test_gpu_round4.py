"""GPU tests of what round 4 changed (run on the MI355X box: pytest -m gpu); everything goes through the C ABI.

  * fp32 fused broadband solvers in the one-column-per-lane geometry (odd column counts, with and without g, all tilings);
  * the persistent index tables of the windowed gas optics are validated against the k-distribution's CONTENTS;
  * the workspace of the any-nlay solver forms belongs to its stream and is handed back;
  * the small kernels that were rewritten (toa source, range sums, gather grid).
"""
import ctypes
import numpy as np
import pytest
import torch

import cases
from rte_rrtmgp_cpp_amd import synthetic, pipeline

pytestmark = pytest.mark.gpu


def _random_columns(rng, ngpt, nlay, ncol):
    tau = 10.0**rng.uniform(-5, 1.5, (ngpt, nlay, ncol)); ssa = rng.uniform(0, 1, tau.shape); g = rng.uniform(0, .9, tau.shape)
    lay = rng.uniform(5, 40, tau.shape); lev = rng.uniform(5, 40, (ngpt, nlay+1, ncol))
    e2 = rng.uniform(.5, 1, (ngpt, ncol)); mu0 = rng.uniform(.1, 1, ncol)
    return tau, ssa, g, lay, lev, e2, mu0


@pytest.mark.parametrize("with_g", [True, False], ids=["g", "no-g"])
@pytest.mark.parametrize("ncol,nlay,top_at_1", [(33, 140, False), (17, 60, True), (129, 20, False), (40, 143, True), (31, 190, False)])
def test_fp32_broadband_solvers_one_column_per_lane(ncol, nlay, top_at_1, with_g, hip_f32, oracle_f32):
    """fp32 do_broadband: the 16 x 4-lane geometry (SW default; LW for odd column counts and under variant 15) against the fp32
    oracle, on odd and even column counts, every K of the tiling (20 ... 190 layers), with g (all-sky form, pipelined loads of tau,
    ssa AND g) and without (clear-sky form). Bounds: twice the worst error observed."""
    rng = np.random.default_rng(100*nlay + ncol)
    ngpt = 6
    tau, ssa, g, lay, lev, e2, mu0 = _random_columns(rng, ngpt, nlay, ncol)
    out = []
    for be, lwv in ((hip_f32, 15), (hip_f32, 0), (oracle_f32, 0)):
        up = be.asarray
        if be is hip_f32:
            be.set_variant(lw=lwv)
        try:
            sec = be.lw_secants_array(ncol, ngpt, 1, 4, up(pipeline.GAUSS_DS)); w = up(np.array([1.0]))
            l = be.lw_solver_noscat(top_at_1, sec, w, up(tau), up(lay), up(lev), up(e2), up(e2*20), inc_flux=up(e2*3), do_broadband=True)
            gg = up(g) if with_g else (None if be is hip_f32 else be.zeros(tau.shape))
            s = be.sw_solver_2stream(top_at_1, up(tau), up(ssa), gg, up(mu0), up(e2*.5), up(e2*.4), up(e2*3), inc_flux_dif=up(e2*.2),
                                     do_broadband=True)
        finally:
            if be is hip_f32:
                be.set_variant(lw=0)
        out.append([be.to_numpy(x) for x in (l["flux_up"], l["flux_dn"], s["flux_up"], s["flux_dn"], s["flux_dir"])])
    for name, a15, a0, o in zip(("lw_up", "lw_dn", "sw_up", "sw_dn", "sw_dir"), *out):
        assert a0.shape == o.shape == (nlay+1, ncol)
        tol = 5e-5 if name.startswith("lw") else 2e-4           # (twice what is observed: 9e-5 in SW)
        assert cases.rel_err(a0, o, floor=1e-2) <= tol, name
        assert cases.rel_err(a15, o, floor=1e-2) <= tol, name + " (one column per lane)"


def test_fp32_sw_geometries_agree(hip_f32):
    """The two fp32 SW tilings (one column per lane, default; two columns per lane, variant 9) on the same inputs."""
    rng = np.random.default_rng(5)
    ngpt, nlay, ncol = 5, 140, 64
    tau, ssa, g, _, _, e2, mu0 = _random_columns(rng, ngpt, nlay, ncol)
    be = hip_f32; up = be.asarray
    res = []
    for v in (0, 9):
        be.set_variant(sw=v)
        try:
            s = be.sw_solver_2stream(False, up(tau), up(ssa), up(g), up(mu0), up(e2*.5), up(e2*.4), up(e2*3), do_broadband=True)
        finally:
            be.set_variant(sw=0)
        res.append([be.to_numpy(s[k]) for k in ("flux_up", "flux_dn", "flux_dir")])
    for a, b in zip(*res):
        assert cases.rel_err(a, b, floor=1e-2) <= 5e-6          # (observed 6e-7)


def test_window_tables_follow_the_contents_not_the_pointers(hip_f64, oracle_f64):
    """The windowed gas optics keeps its index tables per k-distribution between launches and validates them against the index
    arrays' contents at every launch. Overwriting a k-distribution IN PLACE (same device addresses, other flavors / contributor
    intervals) must give the new k-distribution's optical depths."""
    be, orc = hip_f64, oracle_f64
    atm0 = synthetic.make_atmosphere(256, 24, nbnd_lw=4, nbnd_sw=4)
    kds = [synthetic.make_kdist("lw", ngpt=64, nbnd=4, npres=14, nflav=4, nminor_lower=6, nminor_upper=4, seed=s) for s in (3, 4)]
    dev = be.upload_kdist(kds[0])
    atm = pipeline.upload_atmosphere(be, atm0)
    for i, kd0 in enumerate(kds):
        if i > 0:                                   # second k-distribution written over the first one's device arrays
            other = be.upload_kdist(kd0)
            for name, t in vars(dev).items():
                src = getattr(other, name)
                if torch.is_tensor(t):
                    assert t.shape == src.shape, name
                    t.copy_(src)
                elif not isinstance(t, (list, tuple, dict)):
                    setattr(dev, name, src)
            torch.cuda.synchronize()
        got = pipeline.solve_lw(be, dev, atm, do_broadband=True)
        ref = pipeline.solve_lw(orc, orc.upload_kdist(kd0), pipeline.upload_atmosphere(orc, atm0), do_broadband=True)
        for k in ("flux_up", "flux_dn"):
            assert cases.rel_err(be.to_numpy(got[k]), orc.to_numpy(ref[k])) <= 1e-9, (i, k)


def test_workspace_belongs_to_its_stream(hip_f64):
    """the grow-only block of the any-nlay solver forms is owned by the stream: repeated general-path calls reuse ONE
    block (device memory does not grow), rrx_release_workspace / rrx_stream_destroy hand it back, and the fractions entry and the
    general entry it calls share one lease (results equal those of separately provided arrays)."""
    be = hip_f64
    lib = be.lib.cdll
    lib.rrx_workspace_bytes.restype = ctypes.c_ulonglong
    rng = np.random.default_rng(9)
    ngpt, nlay, ncol = 4, 600, 48                      # 600 layers: beyond every fused tiling (575) -> workspace forms
    tau, ssa, g, lay, lev, e2, mu0 = _random_columns(rng, ngpt, nlay, ncol)
    up = be.asarray
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def solve():
        sec = be.lw_secants_array(ncol, ngpt, 1, 4, up(pipeline.GAUSS_DS)); w = up(np.array([1.0]))
        l = be.lw_solver_noscat(False, sec, w, up(tau), up(lay), up(lev), up(e2), up(e2*20), do_broadband=True)
        s = be.sw_solver_2stream(False, up(tau), up(ssa), up(g), up(mu0), up(e2*.5), up(e2*.4), up(e2*3), do_broadband=True)
        return be.to_numpy(l["flux_up"]), be.to_numpy(s["flux_up"])

    first = solve()
    torch.cuda.synchronize()
    size1 = lib.rrx_workspace_bytes(stream)
    assert size1 > 0
    free1 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        again = solve()
    torch.cuda.synchronize()
    assert lib.rrx_workspace_bytes(stream) == size1
    assert torch.cuda.mem_get_info()[0] >= free1 - (8 << 20), "device memory grew across general-path calls"
    for a, b in zip(first, again):
        assert np.array_equal(a, b)
    assert lib.rrx_release_workspace(stream) == 0
    assert lib.rrx_workspace_bytes(stream) == 0
    # a stream of our own: the block goes with the stream
    s2 = ctypes.c_void_p()
    assert lib.rrx_stream_create(ctypes.byref(s2)) == 0
    with torch.cuda.stream(torch.cuda.ExternalStream(s2.value)):
        solve()
        torch.cuda.synchronize()
        assert lib.rrx_workspace_bytes(s2) > 0
    assert lib.rrx_stream_destroy(s2) == 0
    assert lib.rrx_workspace_bytes(s2) == 0


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_toa_source_equals_spread_then_scale(dt, hip_f64, hip_f32):
    be = hip_f64 if dt == "f64" else hip_f32
    rng = np.random.default_rng(2)
    for ncol, ngpt in ((1, 1), (257, 7), (2048, 224)):
        src = be.asarray(rng.uniform(0.1, 5, ngpt)); tsi = be.asarray(rng.uniform(0.3, 1.2, ncol))
        a = be.spread_col(ncol, src); be.scaling_to_subset(a, tsi)
        b = be.toa_source(ncol, src, tsi)
        assert np.array_equal(be.to_numpy(a), be.to_numpy(b))
        assert np.array_equal(be.to_numpy(a), (be.to_numpy(src)[:, None] * be.to_numpy(tsi)[None, :]).astype(be.np_dtype))


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_cloud_optics_with_delta_scaling_in_one_pass(dt, hip_f64, hip_f32):
    """rrx_cloud_optics_2str_delta = rrx_cloud_optics_2str followed by rrx_delta_scale_2str_k (Radiation_solver.cu:773-792), bit for bit."""
    be = hip_f64 if dt == "f64" else hip_f32
    nbnd = 5
    atm0 = synthetic.make_atmosphere(96, 30, nbnd_lw=nbnd, nbnd_sw=nbnd, clouds=True, seed=4)
    lut = be.upload_lut({k: (v.astype(be.np_dtype) if isinstance(v, np.ndarray) else v) for k, v in synthetic.make_cloud_lut(nbnd, "sw").items()})
    atm = pipeline.upload_atmosphere(be, atm0.astype(be.np_dtype))
    two = be.cloud_optics_2str(lut, atm.lwp, atm.iwp, atm.rel, atm.dei)
    be.delta_scale_2str_k(*two)
    one = be.cloud_optics_2str(lut, atm.lwp, atm.iwp, atm.rel, atm.dei, delta_scale=True)
    assert float(be.to_numpy(two[0]).max()) > 0
    for a, b in zip(two, one):
        assert np.array_equal(be.to_numpy(a), be.to_numpy(b))


@pytest.mark.parametrize("spread", [0.0, 0.35], ids=["alike", "differing"])
def test_resident_solver_pads_and_orders_columns(spread, hip_f64, monkeypatch):
    """ResidentSolver runs its step on a multiple of 16 columns (padding by repeats of the last column) and, where neighbouring
    columns differ, in order of surface pressure -- one cached gather index for both; the caller sees its own columns in its own
    order. Against the same solver with padding and sorting switched off."""
    be = hip_f64
    ncol, nlay = 273, 40
    kw = dict(ngpt=64, nbnd=4, npres=20, nflav=4, nminor_lower=9, nminor_upper=5)
    kl, ks = be.upload_kdist(synthetic.make_kdist("lw", **kw)), be.upload_kdist(synthetic.make_kdist("sw", **kw))
    atm0 = synthetic.make_atmosphere(ncol, nlay, nbnd_lw=4, nbnd_sw=4, seed=5)
    if spread > 0:
        rng = np.random.default_rng(8)
        f = rng.uniform(1 - spread, 1 + spread, ncol)
        atm0.p_lay = np.ascontiguousarray(atm0.p_lay * f); atm0.p_lev = np.ascontiguousarray(atm0.p_lev * f)
    atm = pipeline.upload_atmosphere(be, atm0)
    monkeypatch.setenv("RRX_PAD_COLUMNS", "0")
    plain = pipeline.ResidentSolver(be, kl, ks, atm, do_broadband=True, sort_columns="0")
    assert plain.perm is None
    ref = be.to_numpy(plain.step()).copy()
    monkeypatch.setenv("RRX_PAD_COLUMNS", "1")
    solver = pipeline.ResidentSolver(be, kl, ks, atm, do_broadband=True, sort_columns="auto")
    assert solver.npad == 15 and solver.perm.numel() == 288 and solver.sort_columns == (spread > 0)
    got = be.to_numpy(solver.step())
    assert got.shape == ref.shape == (7, nlay+1, ncol)
    # (a column meets other neighbours in the sorted run: windowed or gather gas optics, 1e-15 apart, amplified by the SW solver)
    assert cases.rel_err(got[:3], ref[:3]) <= 1e-11 and cases.rel_err(got[3:], ref[3:]) <= 1e-7
    again = be.to_numpy(solver.step())
    assert np.array_equal(got, again)


@pytest.mark.parametrize("clouds", [False, True], ids=["clear", "allsky"])
def test_cxx_host_classes_driven_from_python(clouds, hip_f64):
    """bench.py --driver cxx: Radiation_solver_longwave / _shortwave::solve_gpu (the reference's class structure) through the C entry
    points of cxx_driver_api.cpp on device arrays torch owns, against pipeline.ResidentSolver on the same kernels. 273 columns with
    a pressure spread: the solvers sort and pad the columns on the device and hand the fluxes back in the caller's order."""
    from rte_rrtmgp_cpp_amd import cxx_driver
    be = hip_f64
    ncol, nlay, nbnd = 273, 40, 4
    kw = dict(ngpt=64, nbnd=nbnd, npres=20, nflav=4, nminor_lower=9, nminor_upper=5)
    kl0, ks0 = synthetic.make_kdist("lw", **kw), synthetic.make_kdist("sw", **kw)
    atm0 = synthetic.make_atmosphere(ncol, nlay, nbnd_lw=nbnd, nbnd_sw=nbnd, seed=5, clouds=clouds)
    rng = np.random.default_rng(8)
    f = rng.uniform(0.65, 1.35, ncol)
    atm0.p_lay = np.ascontiguousarray(atm0.p_lay * f); atm0.p_lev = np.ascontiguousarray(atm0.p_lev * f)
    atm = pipeline.upload_atmosphere(be, atm0)
    luts0 = (synthetic.make_cloud_lut(nbnd, "lw"), synthetic.make_cloud_lut(nbnd, "sw")) if clouds else None
    luts = tuple(be.upload_lut(l) for l in luts0) if clouds else None
    ref = be.to_numpy(pipeline.ResidentSolver(be, be.upload_kdist(kl0), be.upload_kdist(ks0), atm, do_broadband=True, cloud_luts=luts).step()).copy()
    drv = cxx_driver.CxxDriver(be, kl0, ks0, atm, luts0, column_block=ncol)
    try:
        got = be.to_numpy(drv.step()).copy()
        again = be.to_numpy(drv.step())
    finally:
        drv.close()
    assert np.array_equal(got, again)
    assert cases.rel_err(got[:3], ref[:3]) <= 1e-11 and cases.rel_err(got[3:], ref[3:]) <= 1e-7


def test_cpu_boundary_broadband_jacobian_is_sized_like_the_fluxes():
    """rte_lw_solver_noscat at the CPU boundary with do_broadband AND do_jacobians: the Jacobian comes back as one (ncol, nlev) array
    -- the g-point sum -- like the fluxes (src/Rte_lw.cpp:181 sizes it so); nothing is written behind it."""
    import cpu_boundary
    from rte_rrtmgp_cpp_amd._ffi import BoolArg
    b = cpu_boundary.HipCpuBoundary(np.float64)
    rng = np.random.default_rng(12)
    ngpt, nlay, ncol = 5, 30, 20
    tau, _, _, lay, lev, e2, _ = _random_columns(rng, ngpt, nlay, ncol)
    sec = np.full((1, ngpt, ncol), 1.66); w = np.array([1.0])
    args = (sec, w, tau, lay, lev, e2, e2*20, None)
    per_g = b.lw_solver_noscat(False, sec, w, tau, lay, lev, e2, e2*20, do_jacobians=True, sfc_src_jac=e2*0.3)
    up = np.zeros((nlay+1, ncol)); dn = np.zeros((nlay+1, ncol)); dummy = np.zeros(1)
    jac = np.full(2*(nlay+1)*ncol, -7.0)                       # the Jacobian and as much again of guard words
    b.lib.call("rte_lw_solver_noscat", ncol, nlay, ngpt, BoolArg(False), 1, *args, dummy, dummy,
               BoolArg(True), up, dn, BoolArg(True), e2*0.3, jac, BoolArg(False), tau, tau)
    n = (nlay+1)*ncol
    assert np.all(jac[n:] == -7.0), "written behind the (ncol, nlev) Jacobian"
    assert cases.rel_err(jac[:n].reshape(nlay+1, ncol), per_g["flux_up_jac"].sum(axis=0)) <= 1e-12
    assert cases.rel_err(up, per_g["flux_up"].sum(axis=0)) <= 1e-12


@pytest.mark.parametrize("kind", ["lw", "sw"])
@pytest.mark.parametrize("geom", ["0", "1"])
def test_window_kernel_geometries_and_chunk_parts(kind, geom, hip_f64, oracle_f64, monkeypatch):
    """The windowed gas optics in both workgroup shapes (RRX_GW_GEOM: 64 columns x 4 layers / 256 columns x 1 layer) on few columns --
    the chunk loop is then shared out over grid.z, and with 8-g-point bands part 0 takes the whole range -- against the oracle, with
    the census of the launch: the path that ran is asserted, not assumed."""
    import os
    monkeypatch.setenv("RRX_GW_GEOM", geom)
    nbnd = 8
    kd0 = synthetic.make_kdist(kind, ngpt=64, nbnd=nbnd, nminor_lower=2*nbnd, nminor_upper=nbnd)
    atm0 = synthetic.make_atmosphere(320, 28, nbnd_lw=nbnd, nbnd_sw=nbnd, seed=23)
    res = []
    for be in (hip_f64, oracle_f64):
        kd = be.upload_kdist(kd0)
        atm = pipeline.upload_atmosphere(be, atm0)
        if be is hip_f64:
            os.environ["RRX_GW_STATS"] = "1"
            be.lib.cdll.rrx_gas_window_stats(None, None, 1)
        try:
            r = (pipeline.solve_lw if kind == "lw" else pipeline.solve_sw)(be, kd, atm, keep=True, do_broadband=True)
        finally:
            os.environ.pop("RRX_GW_STATS", None)
        if be is hip_f64:
            handed, total = ctypes.c_longlong(0), ctypes.c_longlong(0)
            be.lib.cdll.rrx_gas_window_stats(ctypes.byref(handed), ctypes.byref(total), 1)
            # geometry 1: 2 column blocks x 28 layers, geometry 0: 5 x 7 -- times the grid.z parts of a few-column launch
            blocks = 2*28 if geom == "1" else 5*7
            assert total.value > 0 and total.value % blocks == 0 and total.value // blocks in (1, 2, 4), (total.value, blocks)
            # alike columns: at most the workgroups around the tropopause (geometry 0 only) are handed back
            assert handed.value <= (0 if geom == "1" else total.value // 3), (handed.value, total.value)
        res.append({k: be.to_numpy(v) for k, v in r.items() if v is not None and not isinstance(v, dict)})
    h, o = res
    for k in ("tau", "flux_up", "flux_dn", "flux_net"):
        e = cases.rel_err(h[k], o[k])
        assert e <= (1e-7 if (kind == "sw" and "flux" in k) else 1e-9), f"{kind} {k}: {e:.3e}"
