"""bench.py --driver cxx / tests: the C++ host classes (Radiation_solver_longwave / _shortwave::solve_gpu, the reference's class
structure: include_test/Radiation_solver.h) driven from Python on device arrays torch owns, through the C entry points of
rte-rrtmgp-cpp_amd/host/src_test/cxx_driver_api.cpp in librte_rrtmgp_hip[_sp].so. The k-distributions travel as the files the
reference's driver reads (synthetic_files.write_case)."""
import ctypes
import os
import tempfile

import numpy as np

from . import synthetic_files

HERE = os.path.dirname(os.path.abspath(__file__))


class CxxDriver:
    def __init__(self, be, kd_lw0, kd_sw0, atm, cloud_luts0=None, column_block=16384, broadband=True, sort_mode=-1, pad=True):
        import torch
        self.torch, self.be, self.atm = torch, be, atm
        self.f64 = be.np_dtype == np.float64
        path = os.path.join(HERE, "lib", "librte_rrtmgp_hip.so" if self.f64 else "librte_rrtmgp_hip_sp.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build the host layer (make -C rte-rrtmgp-cpp_amd/host [PRECISION=sp])")
        self.lib = ctypes.CDLL(path)
        self.lib.rrx_cxx_driver_create.restype = ctypes.c_void_p
        self.lib.rrx_cxx_driver_error.restype = ctypes.c_char_p
        self.dir = tempfile.mkdtemp(prefix="rrx_cxx_")
        synthetic_files.write_kdist(os.path.join(self.dir, "coefficients_lw.nc"), kd_lw0)
        synthetic_files.write_kdist(os.path.join(self.dir, "coefficients_sw.nc"), kd_sw0)
        if cloud_luts0 is not None:
            synthetic_files.write_cloud_lut(os.path.join(self.dir, "cloud_coefficients_lw.nc"), cloud_luts0[0])
            synthetic_files.write_cloud_lut(os.path.join(self.dir, "cloud_coefficients_sw.nc"), cloud_luts0[1])
        self.clouds = cloud_luts0 is not None
        names = list(atm.vmr.keys())
        arr = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
        self.h = self.lib.rrx_cxx_driver_create(self.dir.encode(), len(names), arr, int(self.clouds), int(bool(atm.top_at_1)))
        if not self.h:
            raise RuntimeError("rrx_cxx_driver_create: " + self.lib.rrx_cxx_driver_error().decode())
        self.h = ctypes.c_void_p(self.h)
        self._check(self.lib.rrx_cxx_driver_settings(self.h, int(column_block), int(broadband), int(sort_mode), int(pad)))
        for n, t in atm.vmr.items():                # (nlay, ncol) tensors = (ncol, nlay) arrays; profiles (nlay,) = (1, nlay)
            n1, n2 = (t.shape[1], t.shape[0]) if t.dim() == 2 else ((1, t.shape[0]) if t.dim() == 1 else (1, 1))
            self._check(self.lib.rrx_cxx_driver_set_gas(self.h, n.encode(), ctypes.c_void_p(t.data_ptr()), n1, n2))
        self.nbnd_lw, self.nbnd_sw = kd_lw0.nbnd, kd_sw0.nbnd
        self.fluxes = be.empty((7, atm.nlay + 1, atm.ncol))
        self.sort_columns = sort_mode

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("cxx driver: " + self.lib.rrx_cxx_driver_error().decode())

    def step(self):
        a = self.atm
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
        st = ctypes.c_void_p(self.torch.cuda.current_stream(self.be.device).cuda_stream)
        cl = (a.lwp, a.iwp, a.rel, a.dei) if self.clouds else (None, None, None, None)
        out7 = (ctypes.c_void_p * 7)(*[self.fluxes[i].data_ptr() for i in range(7)])      # the solvers write into the packed tensor
        self._check(self.lib.rrx_cxx_driver_solve(self.h, a.ncol, a.nlay, self.nbnd_lw, self.nbnd_sw, p(a.p_lay), p(a.p_lev), p(a.t_lay), p(a.t_lev),
                                                  p(a.t_sfc), p(a.emis_sfc), p(a.sfc_alb_dir), p(a.sfc_alb_dif), p(a.tsi_scaling), p(a.mu0),
                                                  p(cl[0]), p(cl[1]), p(cl[2]), p(cl[3]), out7, st))
        return self.fluxes

    def close(self):
        if self.h:
            self.lib.rrx_cxx_driver_destroy(self.h)
            self.h = None
