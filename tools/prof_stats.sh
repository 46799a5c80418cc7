#!/bin/bash
# Run ON the GPU box: bash tools/prof_stats.sh <tag> [bench args]  -> gpurun_out/<tag>/kernel_stats.csv (rocprofv3 --kernel-trace --stats)
TAG=${1:-prof}; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/bench.py --cpu-cols 0 --steps 5 --warmup 2 "$@" > $OUT/kt.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/kt -name "*kernel_trace.csv" -delete
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    name = r["Name"]
    import re; m = re.search(r"::(\w+)<([^>]*)>", name); name = (m.group(1) + "<" + m.group(2) + ">") if m else name[:80]
    print(f'{float(r["AverageNs"])/1e6:8.3f} ms  x{r["Calls"]:>4}  {float(r["Percentage"]):5.1f}%  {name}')
PY
