import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import cases
from rte_rrtmgp_cpp_amd import synthetic, pipeline, hip_kernels
sys.path.insert(0, "oracle"); import oracle_py
hip = hip_kernels.HipKernels(dtype=np.float64); orc = oracle_py.CpuKernels("oracle")
for kind in ("lw", "sw"):
    for ngpt, nbnd in ((320, 20), (48, 3), (16, 1), (7, 1)):
        kd0 = synthetic.make_kdist(kind, ngpt=ngpt, nbnd=nbnd, npres=20, nflav=4, nminor_lower=2*nbnd+1, nminor_upper=nbnd)
        atm0 = synthetic.make_atmosphere(300, 40, nbnd_lw=nbnd, nbnd_sw=nbnd, seed=3)
        res = []
        for be in (hip, orc):
            kd = be.upload_kdist(kd0); atm = pipeline.upload_atmosphere(be, atm0)
            fn = pipeline.solve_lw if kind == "lw" else pipeline.solve_sw
            r = fn(be, kd, atm, keep=True, do_broadband=True)
            res.append({k: be.to_numpy(v) for k, v in r.items() if v is not None and not isinstance(v, dict)})
        h, o = res
        print(kind, ngpt, nbnd, {k: float("%.2e" % cases.rel_err(h[k], o[k])) for k in ("tau", "flux_up", "flux_dn")})
