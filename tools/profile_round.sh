#!/bin/bash
# Round profile, run ON the GPU box from the repo root:  bash tools/profile_round.sh <round tag, e.g. r01>
# Writes gpurun_out/<tag>/: the bench JSON lines (both flux modes, both precisions), rocprofv3 kernel-trace stats of
# the default bench command, and FETCH_SIZE / WRITE_SIZE counter passes (separate passes, no trace domains mixed in).
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p $OUT
for dt in f64 f32; do
  for mode in broadband per-gpoint; do
    timeout -k 10 400 python3 bench.py --dtype $dt --flux-mode $mode > $OUT/bench_${dt}_${mode}.log 2>&1 || echo "bench $dt $mode FAILED"
    grep '^{' $OUT/bench_${dt}_${mode}.log | tail -1 > $OUT/${TAG}_bench_${dt}_${mode}.json
    echo "bench $dt $mode done"
  done
done
cd /tmp
for mode in broadband per-gpoint; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$mode -o kt -- python3 $REPO/bench.py --flux-mode $mode --cpu-cols 0 > $OUT/kt_$mode.log 2>&1 || echo "kernel trace $mode FAILED"
  find $OUT/kt_$mode -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_rocprofv3_kernel_stats_f64_$mode.csv \;
  find $OUT/kt_$mode -name "*kernel_trace.csv" -delete
  echo "kernel trace $mode done"
  for c in FETCH_SIZE WRITE_SIZE SQ; do
    [ $c = SQ ] && CTRS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" || CTRS=$c
    timeout -k 10 400 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_${mode}_$c -o pmc -- python3 $REPO/bench.py --flux-mode $mode --cpu-cols 0 --steps 3 --warmup 1 > $OUT/pmc_${mode}_$c.log 2>&1 || echo "pmc $mode $c FAILED"
    echo "pmc $mode $c done"
  done
done
cd $REPO
for mode in broadband per-gpoint; do
  mkdir -p $OUT/sel_$mode; rm -rf $OUT/sel_$mode/*
  for c in FETCH_SIZE WRITE_SIZE SQ; do cp -r $OUT/pmc_${mode}_$c $OUT/sel_$mode/; done
  python3 tools/pmc_summary.py $OUT/sel_$mode --json $OUT/${TAG}_pmc_traffic.json --tag "f64|$mode|16384x140x256" > $OUT/${TAG}_pmc_${mode}.txt
  rm -rf $OUT/sel_$mode
done
find $OUT -name "*counter_collection.csv" -size +8M -delete
ls -la $OUT | head -40
