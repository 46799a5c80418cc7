#!/bin/bash
# Run ON the GPU box: kernel timeline of the C++ driver at C4 (one LW + one SW solve of the timed runs): busy time, gaps, the
# kernels and copies in order -- where the driver's 0.3-0.5 ms per solve above the sum of its big kernels goes.
OUT=$PWD/gpurun_out/drvgap; mkdir -p $OUT; REPO=$PWD
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
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/kt -o kt -- $REPO/rte-rrtmgp-cpp_amd/lib/test_rte_rrtmgp_gpu --timings --async > $OUT/drv.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"::(\w+)<", r["Kernel_Name"]); name = m.group(1) if m else r["Kernel_Name"][:40]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
for f in glob.glob(out + "/kt/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "")))
ev.sort()
# the last LW solve = from the last gas_window_kernel<2,PF> ... ; take the last 2 windows between fill_gases kernels
idx = [i for i, e in enumerate(ev) if e[2] == "fill_gases_all_kernel"]
for label, a, b in (("second-to-last solve", idx[-2], idx[-1]),):
    seg = ev[a:b]
    t0, t1 = seg[0][0], seg[-1][1]
    busy = sum(e[1] - e[0] for e in seg)
    print(label, "span %.3f ms, busy %.3f ms, %d events" % ((t1 - t0)/1e6, busy/1e6, len(seg)))
    prev = seg[0][0]
    for s, e, n in seg:
        print("  gap %7.1f us  run %8.1f us  %s" % ((s - prev)/1e3, (e - s)/1e3, n)); prev = e
PY
find $OUT/kt -name "*.csv" -size +1M -delete
