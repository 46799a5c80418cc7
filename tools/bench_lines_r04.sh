#!/bin/bash
# The bench JSON lines of profiles/ alone (no profiler), run ON the GPU box: after tools/profile_r04.sh has produced the counters of
# the current sources and profiles/pmc_traffic.json has been replaced by them, the lines carry traffic_stale: false.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04b; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" 2>/dev/null | grep '^{' | tail -1 > $OUT/r04_bench_$name.json; python3 -c "import json; d=json.load(open('$OUT/r04_bench_$name.json')); print('$name', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('traffic_stale'))"; }
run f64_broadband
run f64_per-gpoint --per-gpoint --cpu-cols 0
run f32_broadband --dtype f32 --cpu-cols 0
run f32_per-gpoint --dtype f32 --per-gpoint --cpu-cols 0
run f32_allsky_ncol32768 --dtype f32 --allsky --ncol 32768 --cpu-cols 0
run f64_allsky --allsky --cpu-cols 0
run f64_broadband_ncol2048 --ncol 2048 --cpu-cols 0
run f64_broadband_ncol4096 --ncol 4096 --cpu-cols 0
run f64_broadband_ncol8192 --ncol 8192 --cpu-cols 0
run f64_broadband_driver_cxx --driver cxx --cpu-cols 0
