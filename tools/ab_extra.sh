# A/B of build-time flags on ONE box: bash tools/ab_extra.sh "<name> <EXTRA flags>" ...  (each case: rebuild the solvers, two bench runs)
export TMPDIR=/tmp
for case in "$@"; do
  name=${case%% *}; extra=${case#* }; [ "$extra" = "$name" ] && extra=""
  touch rte-rrtmgp-cpp_amd/csrc/*.hip
  make -C rte-rrtmgp-cpp_amd/csrc EXTRA="$extra" > gpurun_out/abx_build_$name.log 2>&1 || { echo BUILD FAIL $name; exit 1; }
  for i in 1 2; do
    timeout -k 10 200 python bench.py --cpu-cols 0 --steps 10 $BENCH_ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], {k:round(v['ms'],3) for k,v in d['stages'].items() if v['ms']>0.02})"
  done
done
