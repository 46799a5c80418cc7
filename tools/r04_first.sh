#!/bin/bash
# round-4 opening measurements on the GPU box: fp32 issue costs, bench lines of the four configurations, fp32 all-sky kernel stats
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04a; mkdir -p $OUT; REPO=$PWD
timeout -k 10 200 tools/_build/issue_bench > $OUT/issue_costs.txt 2>&1; echo "issue bench rc=$?"
SKIP_TESTS=1 bash tools/gpu_check.sh r04a_bench "" "--dtype f32" "--dtype f32 --allsky --ncol 32768" "--allsky" "--ncol 2048" "--dtype f32 --ncol 32768"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_f32_allsky -o kt -- python3 $REPO/bench.py --dtype f32 --allsky --ncol 32768 --cpu-cols 0 > $OUT/kt_f32_allsky.log 2>&1 || echo "kernel trace FAILED"
find $OUT/kt_f32_allsky -name "*kernel_stats.csv" -exec cp {} $OUT/r04a_rocprofv3_kernel_stats_f32_allsky_ncol32768.csv \;
find $OUT/kt_f32_allsky -name "*kernel_trace.csv" -delete
head -12 $OUT/r04a_rocprofv3_kernel_stats_f32_allsky_ncol32768.csv | cut -c1-220
