# A/B of the working tree's rrx_gas_optics.hip against a saved copy, ONE box, clear-sky and all-sky. Before the call:
#   git show HEAD:rte-rrtmgp-cpp_amd/csrc/rrx_gas_optics.hip > tools/_ab_head_gas_optics.hip   (git-ignored; travels with gpurun)
export TMPDIR=/tmp
cp rte-rrtmgp-cpp_amd/csrc/rrx_gas_optics.hip /tmp/new_gas_optics.hip
for v in new head new head; do
  if [ $v = head ]; then cp tools/_ab_head_gas_optics.hip rte-rrtmgp-cpp_amd/csrc/rrx_gas_optics.hip; else cp /tmp/new_gas_optics.hip rte-rrtmgp-cpp_amd/csrc/rrx_gas_optics.hip; fi
  make -C rte-rrtmgp-cpp_amd/csrc > /tmp/abh_build.log 2>&1 || { echo BUILD FAIL $v; exit 1; }
  for a in "" "--allsky"; do
    timeout -k 10 300 python bench.py --cpu-cols 0 --steps 10 $a 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v [$a]', d['ms_per_step'], {k:round(v['ms'],3) for k,v in d['stages'].items() if v['ms']>0.02})"
  done
done
cp /tmp/new_gas_optics.hip rte-rrtmgp-cpp_amd/csrc/rrx_gas_optics.hip
