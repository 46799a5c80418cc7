#!/bin/bash
# A/B builds of the whole device library ON the GPU box (same box, same session): bash tools/ab_all.sh "<EXTRA A>" "<EXTRA B>" ...
# Each build is benchmarked twice; with TESTS=1 the GPU parity tests run on it too (failures listed, not fatal).
mkdir -p gpurun_out/ab_all
i=0
for ex in "$@"; do
  i=$((i+1))
  touch rte-rrtmgp-cpp_amd/csrc/*.hip
  make -C rte-rrtmgp-cpp_amd/csrc -j4 EXTRA="$ex" > gpurun_out/ab_all/build_$i.log 2>&1 || { echo "build failed: $ex"; tail -5 gpurun_out/ab_all/build_$i.log; continue; }
  echo "== EXTRA=$ex"
  SKIP_TESTS=1 bash tools/gpu_check.sh ab_all_$i "" ""
  if [ "$TESTS" = "1" ]; then
    timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/ab_all/pytest_$i.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/ab_all/pytest_$i.log | cut -c1-160 | tail -25
  fi
done
