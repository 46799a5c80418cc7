#!/usr/bin/env python
"""Micro-benchmark of the two solver kernels alone (random optical properties, C4 shape), for A/B-ing kernel variants
in ONE process (interleaved rounds, cdna_hip_programming.md section 5.4 rule 24). Prints GB/s of algorithmic traffic."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rte_rrtmgp_cpp_amd as R
from rte_rrtmgp_cpp_amd import pipeline

ap = argparse.ArgumentParser()
ap.add_argument("--ncol", type=int, default=16384); ap.add_argument("--nlay", type=int, default=140)
ap.add_argument("--ngpt", type=int, default=256); ap.add_argument("--dtype", default="f64")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--lw", default="0,2,1"); ap.add_argument("--sw", default="0,1")
a = ap.parse_args()
dt = np.float64 if a.dtype == "f64" else np.float32
be = R.HipKernels(dt)
g = torch.Generator(device="cuda").manual_seed(1)
def rnd(shape, lo, hi): return (torch.rand(shape, generator=g, device="cuda", dtype=be.tdtype)*(hi-lo)+lo)
ncol, nlay, ngpt = a.ncol, a.nlay, a.ngpt
tau = 10.0**rnd((ngpt, nlay, ncol), -4, 1); ssa = rnd(tau.shape, 0, 1); gg = rnd(tau.shape, 0, .9)
lay = rnd(tau.shape, 5, 40); lev = rnd((ngpt, nlay+1, ncol), 5, 40)
e2 = rnd((ngpt, ncol), .5, 1); mu0 = rnd((ncol,), .2, 1)
sec = be.lw_secants_array(ncol, ngpt, 1, 4, be.asarray(pipeline.GAUSS_DS)); w = be.asarray(np.array([1.0]))
fu = be.empty((ngpt, nlay+1, ncol)); fd = be.empty((ngpt, nlay+1, ncol)); fr = be.empty((ngpt, nlay+1, ncol))
S = np.dtype(dt).itemsize
lw_bytes = (3*nlay+3+2*(nlay+1))*ncol*ngpt*S; sw_bytes = (3*nlay+3+3*(nlay+1))*ncol*ngpt*S
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
res = {}
for r in range(a.rounds):
    for v in [int(x) for x in a.lw.split(",") if x != ""]:
        be.set_variant(lw=v)
        res.setdefault(("lw", v), []).append(timeit(lambda: be.lw_solver_noscat_into(False, sec, w, tau, lay, lev, e2, e2*20, fu, fd)))
    for v in [int(x) for x in a.sw.split(",") if x != ""]:
        be.set_variant(sw=v)
        res.setdefault(("sw", v), []).append(timeit(lambda: be.sw_solver_2stream_into(False, tau, ssa, gg, mu0, e2*.5, e2*.4, e2*3, fu, fd, fr)))
for (k, v), ts in res.items():
    b = lw_bytes if k == "lw" else sw_bytes
    print(f"{k} variant {v}: median {np.median(ts):8.3f} ms  min {min(ts):8.3f} ms  -> {b/np.median(ts)/1e6:8.1f} GB/s algorithmic ({b/np.median(ts)/1e6/8000*100:5.1f}% of 8 TB/s)")
# calibration kernel for PMC byte counters: tau1 += tau2 reads 2 and writes 1 array of ncol*nlay*ngpt words (8 B/lane)
be.increment_1scalar_by_1scalar(lay, tau); torch.cuda.synchronize()
print("calibration: increment_1scalar_by_1scalar read", 2*tau.numel()*S/1e9, "GB write", tau.numel()*S/1e9, "GB")
