# A/B of build-time flags on reduced k-distribution shapes, ONE box: bash tools/ab_bands.sh "<name> <EXTRA flags>" ...
export TMPDIR=/tmp
for case in "$@"; do
  name=${case%% *}; extra=${case#* }; [ "$extra" = "$name" ] && extra=""
  touch rte-rrtmgp-cpp_amd/csrc/*.hip
  make -C rte-rrtmgp-cpp_amd/csrc EXTRA="$extra" > gpurun_out/abb_build_$name.log 2>&1 || { echo BUILD FAIL $name; exit 1; }
  for a in "--ngpt 128 --nbnd 16" "--ngpt 112 --nbnd 14" "--ngpt 256"; do
    timeout -k 10 300 python bench.py --cpu-cols 0 --steps 10 $a 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name [$a]', d['ms_per_step'], d['gas_window']['handed_back'], {k:round(v['ms'],2) for k,v in d['stages'].items() if v['ms']>0.02})"
  done
done
