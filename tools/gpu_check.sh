#!/bin/bash
# Run ON the GPU box from the repo root: bash tools/gpu_check.sh <tag> [bench variants...]
# GPU test suite + bench lines (stage times printed compactly). Logs under gpurun_out/<tag>/.
TAG=${1:-chk}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ "$SKIP_TESTS" != "1" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
fi
i=0
run_bench() {
  i=$((i+1))
  timeout -k 10 300 python bench.py --cpu-cols 0 "$@" > $OUT/bench_$i.log 2>&1
  python - "$OUT/bench_$i.log" "$*" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"[{sys.argv[2]}] {d['value']:.0f} col/s {d['ms_per_step']:.2f} ms/step", {k: round(v["ms"], 2) for k, v in d["stages"].items()})
except Exception as e:
    print("bench failed:", sys.argv[2], e); print(open(sys.argv[1]).read()[-1500:])
PY
}
if [ $# -eq 0 ]; then run_bench; else for v in "$@"; do run_bench $v; done; fi
