#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs by kernel (mean counter value per dispatch).

  pmc_summary.py <dir> [--json out.json --tag TAG]

With --json, FETCH_SIZE / WRITE_SIZE (rocprofv3 reports KB) become HBM-side bytes per launch with the gfx950 correction of
MI355X_MICROARCH.md section HBM (FETCH_SIZE doubled, WRITE_SIZE as is) and are merged into out.json under
"<TAG>|<kernel short name>"; bench.py reads that file to fill roofline.traffic."""
import argparse, collections, csv, glob, json, os, re

ap = argparse.ArgumentParser()
ap.add_argument("dir"); ap.add_argument("--json"); ap.add_argument("--tag", default="")
ap.add_argument("--filter", default="")
a = ap.parse_args()

def short(name):
    m = re.search(r"::(\w+)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2)}>" if m else name.split("(")[0][:60]

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(a.dir + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if a.filter and a.filter not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:44s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
if a.json:
    out = json.load(open(a.json)) if os.path.exists(a.json) else {}
    for k, cs in acc.items():
        e = out.setdefault(f"{a.tag}|{k}", {})
        if "FETCH_SIZE" in cs:
            e["fetch_bytes"] = 2.0 * 1024.0 * sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
        if "WRITE_SIZE" in cs:
            e["write_bytes"] = 1024.0 * sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        if "SQ_INSTS_VALU" in cs:          # wave-level VALU instructions per launch (bench.py: roofline_valu)
            e["insts_valu"] = sum(cs["SQ_INSTS_VALU"]) / len(cs["SQ_INSTS_VALU"])
    # digest of the device sources the counters belong to (bench.py marks roofline.traffic / roofline_valu stale when it differs)
    import hashlib
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(here, "rte-rrtmgp-cpp_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    out["sources_digest"] = {"sha256": h.hexdigest()}
    json.dump(out, open(a.json, "w"), indent=1, sort_keys=True)
