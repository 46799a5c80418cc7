#!/bin/bash
# Profile ONE bench configuration on the GPU box, from the repo root:
#   bash tools/profile_config.sh <round tag> <name> <pmc tag: dtype|mode|NCOLxNLAYxNGPT> <bench.py arguments...>
# e.g. bash tools/profile_config.sh r04 f32_allsky_ncol32768 "f32|broadband-allsky|32768x140x256" --dtype f32 --allsky --ncol 32768
# Writes gpurun_out/<tag>/: the bench JSON line, the rocprofv3 kernel statistics of the same command, and FETCH_SIZE / WRITE_SIZE /
# SQ counter passes (separate passes, no trace domain mixed in) summarised by tools/pmc_summary.py into <tag>_pmc_<name>.txt and
# <tag>_pmc_traffic.json (merge that into profiles/pmc_traffic.json; bench.py reads roofline.traffic / roofline_valu from it).
set -o pipefail
export TMPDIR=/tmp
TAG=$1; NAME=$2; PTAG=$3; shift 3
OUT=$PWD/gpurun_out/$TAG; REPO=$PWD
mkdir -p $OUT
timeout -k 10 400 python3 bench.py "$@" > $OUT/bench_$NAME.log 2>&1 || echo "bench $NAME FAILED"
grep '^{' $OUT/bench_$NAME.log | tail -1 > $OUT/${TAG}_bench_$NAME.json
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$NAME -o kt -- python3 $REPO/bench.py "$@" --cpu-cols 0 > $OUT/kt_$NAME.log 2>&1 || echo "kernel trace $NAME FAILED"
find $OUT/kt_$NAME -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_rocprofv3_kernel_stats_$NAME.csv \;
find $OUT/kt_$NAME -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE SQ; do
  [ $c = SQ ] && CTRS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" || CTRS=$c
  timeout -k 10 400 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_${NAME}_$c -o pmc -- python3 $REPO/bench.py "$@" --cpu-cols 0 --steps 3 --warmup 1 > $OUT/pmc_${NAME}_$c.log 2>&1 || echo "pmc $NAME $c FAILED"
done
cd $REPO
mkdir -p $OUT/sel_$NAME; rm -rf $OUT/sel_$NAME/*
for c in FETCH_SIZE WRITE_SIZE SQ; do cp -r $OUT/pmc_${NAME}_$c $OUT/sel_$NAME/; done
python3 tools/pmc_summary.py $OUT/sel_$NAME --json $OUT/${TAG}_pmc_traffic.json --tag "$PTAG" > $OUT/${TAG}_pmc_$NAME.txt
rm -rf $OUT/sel_$NAME
find $OUT -name "*counter_collection.csv" -size +8M -delete
echo "profile $NAME done"; grep -A8 -E "sw_2stream_scan_kernel|lw_noscat_bb_kernel" $OUT/${TAG}_pmc_$NAME.txt | head -60
