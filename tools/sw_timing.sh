#!/bin/bash
# Phase clocks of the fused SW and LW solvers (diagnostic build -DRRX_SW_TIMING=1 -DRRX_LW_TIMING=1): bash tools/sw_timing.sh
# prints clocks per (workgroup, g-point) of wavefront 0: two-stream | direct beam | albedo | source | down scan | final replay | barrier waits | loop top
export TMPDIR=/tmp
touch rte-rrtmgp-cpp_amd/csrc/rrx_solver_sw.hip rte-rrtmgp-cpp_amd/csrc/rrx_solver_lw.hip
make -C rte-rrtmgp-cpp_amd/csrc EXTRA="-DRRX_SW_TIMING=1 -DRRX_LW_TIMING=1" > gpurun_out/swt_build.log 2>&1 || { echo BUILD FAIL; tail -5 gpurun_out/swt_build.log; exit 1; }
for args in "" "--dtype f32 --allsky --ncol 32768" "--dtype f32"; do
python3 - $args <<'PY'
import sys, ctypes, subprocess, json, os
sys.path.insert(0, os.getcwd())
import numpy as np
args = sys.argv[1:]
# run bench in-process-like: simplest is to call bench main with the timing read-out afterwards
import torch, rte_rrtmgp_cpp_amd as R
from rte_rrtmgp_cpp_amd import synthetic, pipeline
dt = np.float32 if "f32" in args else np.float64
ncol = int(args[args.index("--ncol")+1]) if "--ncol" in args else 16384
allsky = "--allsky" in args
be = R.HipKernels(dt, "cuda:0")
kl, ks = be.upload_kdist(synthetic.make_kdist("lw", ngpt=256, nbnd=16)), be.upload_kdist(synthetic.make_kdist("sw", ngpt=256, nbnd=16))
atm = pipeline.upload_atmosphere(be, synthetic.make_atmosphere(ncol, 140, nbnd_lw=16, nbnd_sw=16, seed=1234, clouds=allsky).astype(dt))
luts = None
if allsky:
    cast = lambda lut: be.upload_lut({k: (v.astype(dt) if isinstance(v, np.ndarray) else v) for k, v in lut.items()})
    luts = (cast(synthetic.make_cloud_lut(16, "lw")), cast(synthetic.make_cloud_lut(16, "sw")))
s = pipeline.ResidentSolver(be, kl, ks, atm, do_broadband=True, cloud_luts=luts)
for _ in range(2): s.step()
out = (ctypes.c_ulonglong * 128)()
out_lw = (ctypes.c_ulonglong * 128)()
be.lib.cdll.rrx_sw_timing(out); be.lib.cdll.rrx_lw_timing(out_lw)
n = 3
for _ in range(n): s.step()
be.lib.cdll.rrx_sw_timing(out); be.lib.cdll.rrx_lw_timing(out_lw)
V = 1; groups_per_wg = 2
ncol_per_wg = 32 if dt == np.float32 else 16
nwg = (ncol + ncol_per_wg - 1)//ncol_per_wg
names = ["two-stream", "direct beam", "albedo", "source", "down scan", "final replay", "barrier waits", "loop top"]
print(" ".join(args) or "fp64 C4", "| clocks per (workgroup, g-point) of each wavefront of a workgroup:")
for w in range(16):
    per = [out[8*w+k]/(n*nwg*256) for k in range(8)]
    if sum(per) > 0:
        print(f"   wave {w}:", ", ".join(f"{a} {b:.0f}" for a, b in zip(names, per)), "| sum", f"{sum(per):.0f}")
names_lw = ["sources + transmissivities", "down scan", "up scan", "replays + sums", "-", "-", "barrier waits", "loop top"]
nwg_lw = (ncol + 31)//32 if dt == np.float32 else (ncol + 15)//16
print("  LW solver:")
for w in range(16):
    per = [out_lw[8*w+k]/(n*nwg_lw*256) for k in range(8)]
    if sum(per) > 0:
        print(f"   wave {w}:", ", ".join(f"{a} {b:.0f}" for a, b in zip(names_lw, per) if a != "-"), "| sum", f"{sum(per):.0f}")
PY
done
