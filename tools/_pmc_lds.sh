# LDS / VALU / VMEM activity of the windowed gas optics (fp32 and fp64): two passes each
set -o pipefail
export TMPDIR=/tmp
REPO=$PWD; OUT=$PWD/gpurun_out/ldspmc; mkdir -p $OUT
cd /tmp
for dt in f32 f64; do
  timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/a_$dt -o pmc -- python3 $REPO/bench.py --dtype $dt --cpu-cols 0 --steps 2 --warmup 1 > $OUT/a_$dt.log 2>&1 || echo "pass a $dt FAILED"
  timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR --output-format csv -d $OUT/b_$dt -o pmc -- python3 $REPO/bench.py --dtype $dt --cpu-cols 0 --steps 2 --warmup 1 > $OUT/b_$dt.log 2>&1 || echo "pass b $dt FAILED"
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for dt in ("f32", "f64"):
    for p in ("a", "b"):
        for f in glob.glob(f"gpurun_out/ldspmc/{p}_{dt}/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "gas_window_kernel" not in k and "scan_kernel" not in k and "bb_kernel" not in k: continue
                k = k.split("(")[0][-60:]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
            for k, d in acc.items():
                print(dt, p, k, {c: "%.3g" % (v / n[(k, c)]) for c, v in d.items()})
PY
find gpurun_out/ldspmc -name "*counter_collection.csv" -size +4M -delete
