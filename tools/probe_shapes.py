"""Run ON the GPU box: HIP path against the CPU oracle on shapes beside the tested ones (spectral shapes, tall columns, all-sky,
both orientations, fp32): prints the worst relative differences. python tools/probe_shapes.py"""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np
import cases, oracle_py
from rte_rrtmgp_cpp_amd import synthetic, pipeline, hip_kernels

def run(dtype, kind, ngpt, nbnd, ncol, nlay, clouds, top_at_1, spread=0.0):
    hip = hip_kernels.HipKernels(dtype=dtype); orc = oracle_py.CpuKernels("oracle", dtype=dtype)
    kd0 = synthetic.make_kdist(kind, ngpt=ngpt, nbnd=nbnd, npres=20, nflav=4, nminor_lower=2*nbnd+1, nminor_upper=nbnd)
    atm0 = synthetic.make_atmosphere(ncol, nlay, nbnd_lw=nbnd, nbnd_sw=nbnd, seed=3, clouds=clouds, top_at_1=top_at_1)
    if spread:
        rng = np.random.default_rng(5); f = rng.uniform(1-spread, 1+spread, ncol)
        atm0.p_lay = atm0.p_lay*f; atm0.p_lev = atm0.p_lev*f
    lut0 = synthetic.make_cloud_lut(nbnd, kind) if clouds else None
    res = []
    for be in (hip, orc):
        kd = be.upload_kdist(kd0); atm = pipeline.upload_atmosphere(be, atm0.astype(be.np_dtype))
        fn = pipeline.solve_lw if kind == "lw" else pipeline.solve_sw
        kw = dict(cloud_lut=be.upload_lut(lut0)) if clouds else {}
        if clouds and kind == "sw": kw["delta_cloud"] = True
        r = fn(be, kd, atm, keep=True, do_broadband=True, **kw)
        res.append({k: be.to_numpy(v) for k, v in r.items() if v is not None and not isinstance(v, dict)})
    h, o = res
    fl = 1e-2 if dtype == np.float32 else 1e-6
    return {k: float("%.1e" % cases.rel_err(h[k], o[k], floor=fl)) for k in ("tau", "flux_up", "flux_dn")}

for dtype in (np.float64, np.float32):
    for kind in ("lw", "sw"):
        for (ngpt, nbnd, ncol, nlay, clouds, top, spread) in ((320, 20, 300, 40, False, False, 0), (7, 1, 300, 40, False, True, 0),
                (128, 16, 270, 200, True, False, 0), (64, 8, 530, 150, True, True, 0.3), (48, 3, 33, 287, True, False, 0.1)):
            print(dtype.__name__, kind, f"{ngpt}g/{nbnd}b {ncol}x{nlay} clouds={clouds} top_at_1={top} spread={spread}:",
                  run(dtype, kind, ngpt, nbnd, ncol, nlay, clouds, top, spread), flush=True)
