export TMPDIR=/tmp
for sh in 4 1 8; do
  touch rte-rrtmgp-cpp_amd/csrc/rrx_gas_optics.hip
  make -C rte-rrtmgp-cpp_amd/csrc EXTRA="-DRRX_GATHER_SHARES=$sh" > gpurun_out/sp_build_$sh.log 2>&1 || { echo BUILD FAIL; exit 1; }
  for s in 0 0.05 0.35; do
    timeout -k 10 200 python bench.py --cpu-cols 0 --steps 10 --col-spread $s --sort-columns 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('shares $sh spread $s:', d['ms_per_step'], d['gas_window']['handed_back'], {k:round(v['ms'],2) for k,v in d['stages'].items() if v['ms']>0.02})"
  done
done
