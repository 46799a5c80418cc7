export TMPDIR=/tmp
mkdir -p gpurun_out
BENCH_ARGS="--dtype f32" bash tools/gw_timing.sh "f32clear" && BENCH_ARGS="--dtype f32 --allsky --ncol 32768" bash tools/gw_timing.sh "f32allsky" && BENCH_ARGS="" bash tools/gw_timing.sh "f64clear"
