#!/bin/bash
# A/B on the GPU box: fp64 LW fused solver, six waves x six layers per column group (three waves per SIMD) against the product form
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { timeout -k 10 200 python bench.py --cpu-cols 0 --lw-variant $1 > gpurun_out/ab_lw.log 2>&1; tail -1 gpurun_out/ab_lw.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2 lw-variant $1', d['ms_per_step'], {k:round(v['ms'],2) for k,v in d['stages'].items() if v['ms']>0.1})" || tail -5 gpurun_out/ab_lw.log; }
run 0 "EV=2"; run 16 "EV=2"; run 17 "EV=2"
touch rte-rrtmgp-cpp_amd/csrc/rrx_solver_lw.hip && make -C rte-rrtmgp-cpp_amd/csrc EXTRA="-DRRX_LW_EV=1" > gpurun_out/ab_lw_build.log 2>&1 || { echo BUILD FAIL; exit 1; }
run 0 "EV=1"; run 16 "EV=1"; run 17 "EV=1"
