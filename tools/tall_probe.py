import sys, time, importlib
sys.path.insert(0, ".")
import numpy as np, torch
pkg = importlib.import_module("rte_rrtmgp_cpp_amd")
from rte_rrtmgp_cpp_amd import hip_kernels
be = hip_kernels.HipKernels(dtype=np.float64)
ncol, nlay, ngpt = 4096, 288, 256
dev = be.device
tau = torch.rand(ngpt, nlay, ncol, dtype=torch.float64, device=dev)*0.1
ssa = torch.rand(ngpt, nlay, ncol, dtype=torch.float64, device=dev)*0.9
mu0 = torch.full((ncol,), 0.8, dtype=torch.float64, device=dev)
alb = torch.full((ngpt, ncol), 0.1, dtype=torch.float64, device=dev)
inc = torch.ones(ngpt, ncol, dtype=torch.float64, device=dev)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    r = be.sw_solver_2stream(False, tau, ssa, None, mu0, alb, alb, inc, do_broadband=True)
    torch.cuda.synchronize(); print("sw call", i, round((time.time()-t0)*1e3, 1), "ms")
