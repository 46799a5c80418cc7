# Phase clocks of the windowed gas-optics kernel (diagnostic build RRX_GW_TIMING=1): [BENCH_ARGS=..] bash tools/gw_timing.sh "<name> <EXTRA flags>" ...
export TMPDIR=/tmp
for case in "$@"; do
  name=${case%% *}; extra=${case#* }; [ "$extra" = "$name" ] && extra=""
  touch rte-rrtmgp-cpp_amd/csrc/*.hip
  make -C rte-rrtmgp-cpp_amd/csrc EXTRA="-DRRX_GW_TIMING=1 $extra" > gpurun_out/gwt_build_$name.log 2>&1 || { echo BUILD FAIL $name; exit 1; }
  echo "== $name"
  RRX_GW_STATS=1 timeout -k 10 200 python bench.py --cpu-cols 0 --steps 2 --warmup 1 $BENCH_ARGS 2>&1 >/dev/null | grep "clocks per" | tail -2
done
