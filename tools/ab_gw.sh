#!/bin/bash
# A/B builds of the windowed gas-optics kernel ON the GPU box: bash tools/ab_gw.sh "<EXTRA flags A>" "<EXTRA flags B>" ...
mkdir -p gpurun_out/ab_gw
i=0
for ex in "$@"; do
  i=$((i+1))
  touch rte-rrtmgp-cpp_amd/csrc/rrx_gas_optics.hip
  make -C rte-rrtmgp-cpp_amd/csrc EXTRA="$ex" > gpurun_out/ab_gw/build_$i.log 2>&1 || { echo "build failed: $ex"; tail -5 gpurun_out/ab_gw/build_$i.log; continue; }
  echo "== EXTRA=$ex"
  SKIP_TESTS=1 bash tools/gpu_check.sh ab_gw_$i "" 
done
