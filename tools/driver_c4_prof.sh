#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel statistics of the C++ driver on the C4 case (tools/driver_c4.py writes the case)
OUT=$PWD/gpurun_out/drv; mkdir -p $OUT; REPO=$PWD
python3 - <<PY
import os, sys
sys.path.insert(0, "$REPO")
from rte_rrtmgp_cpp_amd import synthetic, synthetic_files
d = "$OUT/case"; os.makedirs(d, exist_ok=True)
kl = synthetic.make_kdist("lw", ngpt=256, nbnd=16); ks = synthetic.make_kdist("sw", ngpt=256, nbnd=16)
atm = synthetic.make_atmosphere(16384, 140, nbnd_lw=16, nbnd_sw=16, seed=1234)
synthetic_files.write_case(d, atm, kl, ks)
PY
cd $OUT/case && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $REPO/rte-rrtmgp-cpp_amd/lib/test_rte_rrtmgp_gpu --timings --async > $OUT/drv.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/kt -name "*kernel_trace.csv" -delete
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    name = r["Name"]; m = re.search(r"::(\w+)<([^>]*)>", name); name = (m.group(1) + "<" + m.group(2) + ">") if m else name[:80]
    print(f'{float(r["AverageNs"])/1e6:8.3f} ms  x{r["Calls"]:>4}  {float(r["Percentage"]):5.1f}%  {name}')
PY
grep Duration $OUT/drv.log | tail -2
