#!/usr/bin/env python
"""The C++ driver (test_rte_rrtmgp_gpu --timings) on the benchmark workload: writes the C4 case (16 384 columns x 140 layers,
256 g-points / 16 bands, synthetic k-distributions of the real shapes) into a scratch directory and runs the stand-alone driver
on it, so that its "Duration ... solver" lines can be put next to bench.py's stage times (same kernels, host classes instead
of the Python pipeline)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rte_rrtmgp_cpp_amd import synthetic, synthetic_files

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
flags = sys.argv[2:] or ["--timings", "--async"]
d = tempfile.mkdtemp(prefix="rrx_c4_")
kl = synthetic.make_kdist("lw", ngpt=256, nbnd=16)
ks = synthetic.make_kdist("sw", ngpt=256, nbnd=16)
atm = synthetic.make_atmosphere(ncol, 140, nbnd_lw=16, nbnd_sw=16, seed=1234)
if os.environ.get("RRX_COL_SPREAD"):                      # the columns of `bench.py --col-spread s`
    import numpy as np
    sp = float(os.environ["RRX_COL_SPREAD"])
    rng = np.random.default_rng(4321)
    scale = rng.uniform(1. - sp, 1. + sp, ncol); dT = rng.uniform(-30.*sp, 30.*sp, ncol)
    atm.p_lay = atm.p_lay * scale; atm.p_lev = atm.p_lev * scale
    atm.t_lay = atm.t_lay + dT; atm.t_lev = atm.t_lev + dT; atm.t_sfc = atm.t_sfc + dT
synthetic_files.write_case(d, atm, kl, ks)
exe = os.path.join(ROOT, "rte-rrtmgp-cpp_amd", "lib", "test_rte_rrtmgp_gpu")
p = subprocess.run([exe] + flags, cwd=d, capture_output=True, text=True)
print("\n".join(l for l in p.stdout.splitlines() if "Duration" in l or "EXCEPTION" in l or "order of surface" in l))
print("exit", p.returncode, p.stderr[-500:])
