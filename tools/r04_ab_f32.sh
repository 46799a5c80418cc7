#!/bin/bash
# A/B on the GPU box: fp32 SW solver geometry (waves per SIMD), clear sky at C4 and all-sky at 32768 columns
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== old two-columns-per-lane form (variant 9)"
for a in "" "--allsky --ncol 32768"; do
  timeout -k 10 200 python bench.py --cpu-cols 0 --dtype f32 --sw-variant 9 $a > gpurun_out/ab_old.log 2>&1
  tail -1 gpurun_out/ab_old.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('old f32 $a', d['ms_per_step'], {k:round(v['ms'],2) for k,v in d['stages'].items() if v['ms']>0.1})"
done
DTYPES=f32 BENCH_ARGS="" bash tools/ab_build.sh rrx_solver_sw.hip < tools/ab_cases_f32_waves.txt
DTYPES=f32 BENCH_ARGS="--allsky --ncol 32768" bash tools/ab_build.sh rrx_solver_sw.hip < tools/ab_cases_f32_waves.txt
