# A/B of build-time tunables of the fused broadband SW kernel (run on the GPU box: bash tools/ab_sw.sh)
export TMPDIR=/tmp
run() { # name, EXTRA
  touch rte-rrtmgp-cpp_amd/csrc/rrx_solver_sw.hip
  make -C rte-rrtmgp-cpp_amd/csrc EXTRA="$2" > gpurun_out/ab_build_$1.log 2>&1 || { echo BUILD FAIL $1; return 1; }
  for dt in $DTYPES; do
  timeout -k 10 200 python bench.py --cpu-cols 0 --broadband --dtype $dt > gpurun_out/ab_$1_bb_$dt.log 2>&1; tail -1 gpurun_out/ab_$1_bb_$dt.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 BB $dt', d['value'], d['ms_per_step'], d['stages']['sw_solver']['ms'], d['stages']['lw_solver']['ms'])"
  done
}
DTYPES=${DTYPES:-"f64 f32"}
while read -r name extra; do [ -n "$name" ] && { run "$name" "$extra" || exit 1; }; done
