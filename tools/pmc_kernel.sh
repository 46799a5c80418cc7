#!/bin/bash
# Run ON the GPU box: bash tools/pmc_kernel.sh <tag> "<counters>" [bench args] -> per-kernel mean of each counter (one rocprofv3 --pmc pass)
TAG=$1; CTR=$2; shift; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc -o pmc -- python3 $REPO/bench.py --cpu-cols 0 --steps 2 --warmup 1 "$@" > $OUT/pmc.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'::(\w+)<([^>]*)>', r["Kernel_Name"])
        name = (m.group(1) + "<" + m.group(2) + ">") if m else r["Kernel_Name"][:60]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in acc.items():
    if not any(k in name for k in ("gas_window", "tau_absorption", "planck", "scan_kernel", "bb_kernel", "bb3_kernel")): continue
    print(name, {k: round(sum(v)/len(v)) for k, v in d.items()})
PY
find $OUT/pmc -name "*.csv" -size +4M -delete
