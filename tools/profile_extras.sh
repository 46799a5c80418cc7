#!/bin/bash
# Run ON the GPU box after tools/profile_round.sh <tag>: the side tables of profiles/README.md -- column spread with and without
# sorting, fewer columns per GPU, the all-sky flow, the C++ driver at C4 (tools/driver_c4.py). Writes gpurun_out/<tag>/.
export TMPDIR=/tmp
TAG=${1:-r01}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms  ', round(d['value']), d['unit'], ' hand-backs', d.get('gas_window'), ' sorted', d['config'].get('columns_sorted'), {k:round(v['ms'],2) for k,v in d['stages'].items() if v['ms']>0.02})"; }
{
for s in 0 0.05 0.35; do for so in 0 auto; do
  timeout -k 10 200 python3 bench.py --cpu-cols 0 --steps 10 --col-spread $s --sort-columns $so 2>/dev/null | line "col-spread $s sort-columns $so:"
done; done
} > $OUT/${TAG}_col_spread.txt
for n in 2048 4096 8192; do
  timeout -k 10 200 python3 bench.py --cpu-cols 0 --ncol $n 2>/dev/null | grep '^{' | tail -1 > $OUT/${TAG}_bench_f64_broadband_ncol$n.json
done
timeout -k 10 300 python3 bench.py --cpu-cols 0 --allsky 2>/dev/null | grep '^{' | tail -1 > $OUT/${TAG}_bench_f64_allsky.json
timeout -k 10 300 python3 bench.py --cpu-cols 0 --allsky --dtype f32 --ncol 32768 2>/dev/null | grep '^{' | tail -1 > $OUT/${TAG}_bench_f32_allsky_ncol32768.json
{
for c in "0 --timings --async" "0.35 --timings --async" "0.35 --timings --async --no-sort-columns"; do
  set -- $c; sp=$1; shift
  echo "== C++ driver (tools/driver_c4.py 16384), col-spread $sp, flags: $@"
  RRX_COL_SPREAD=$sp timeout -k 10 250 python3 tools/driver_c4.py 16384 "$@" | grep -v "^exit 0"
done
} > $OUT/${TAG}_driver_c4_timings.txt 2>&1
ls -la $OUT | grep ${TAG}_ | head -40
