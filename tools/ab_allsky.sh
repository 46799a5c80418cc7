# A/B of build-time flags on the all-sky flow, ONE box: bash tools/ab_allsky.sh "<name> <EXTRA flags>" ...
export TMPDIR=/tmp
for case in "$@"; do
  name=${case%% *}; extra=${case#* }; [ "$extra" = "$name" ] && extra=""
  touch rte-rrtmgp-cpp_amd/csrc/*.hip
  make -C rte-rrtmgp-cpp_amd/csrc EXTRA="$extra" > gpurun_out/aba_build_$name.log 2>&1 || { echo BUILD FAIL $name; exit 1; }
  for a in "--allsky" "--allsky --dtype f32 --ncol 32768"; do
    timeout -k 10 300 python bench.py --cpu-cols 0 --steps 10 $a 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name [$a]', d['ms_per_step'], {k:round(v['ms'],2) for k,v in d['stages'].items() if v['ms']>0.02})"
  done
done
