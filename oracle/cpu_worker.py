"""TEST INFRASTRUCTURE ONLY (CPU baseline leg of bench.py): one worker process of the all-cores CPU baseline.

Solves a share of the 12-column blocks of bench.py's CPU sample with the oracle (the scalar CPU restatement of the
reference CPU path) and returns the time spent in the solves. Imported in spawned processes that never touch the GPU."""
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)
sys.path.insert(0, os.path.dirname(_HERE))


def solve_blocks(job):
    nlay, ngpt, nbnd, starts, ncols = job
    import oracle_py
    from rte_rrtmgp_cpp_amd import synthetic, pipeline
    orc = oracle_py.CpuKernels("oracle", np.float64)
    kd_lw = orc.upload_kdist(synthetic.make_kdist("lw", ngpt=ngpt, nbnd=nbnd))
    kd_sw = orc.upload_kdist(synthetic.make_kdist("sw", ngpt=ngpt, nbnd=nbnd))
    blocks = [synthetic.make_atmosphere(min(12, ncols - c0), nlay, nbnd_lw=nbnd, nbnd_sw=nbnd, seed=1234 + c0) for c0 in starts]
    t0 = time.perf_counter()
    for sub in blocks:
        pipeline.solve_lw(orc, kd_lw, sub, do_broadband=True)
        pipeline.solve_sw(orc, kd_sw, sub, do_broadband=True, fused_gas=False)
    return time.perf_counter() - t0
