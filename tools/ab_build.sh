# Generic A/B of build-time macros on the GPU box:  [DTYPES=..] [BENCH_ARGS=..] bash tools/ab_build.sh <file.hip> < cases.txt   (case line: name EXTRA...)
export TMPDIR=/tmp
SRC=$1
DTYPES=${DTYPES:-"f64 f32"}
while read -r name extra; do
  [ -z "$name" ] && continue
  touch rte-rrtmgp-cpp_amd/csrc/$SRC
  make -C rte-rrtmgp-cpp_amd/csrc EXTRA="$extra" > gpurun_out/ab_build_$name.log 2>&1 || { echo BUILD FAIL $name; exit 1; }
  for dt in $DTYPES; do
    timeout -k 10 200 python bench.py --cpu-cols 0 --dtype $dt $BENCH_ARGS > gpurun_out/ab_${name}_$dt.log 2>&1
    tail -1 gpurun_out/ab_${name}_$dt.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name $dt', d['value'], d['ms_per_step'], {k:round(v['ms'],2) for k,v in d['stages'].items() if v['ms']>0.1})"
  done
done
