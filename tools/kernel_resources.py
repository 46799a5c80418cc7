#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy table of every kernel instantiation of the device library, from the compiler's own
remarks (-Rpass-analysis=kernel-resource-usage, which csrc/Makefile passes and keeps as lib/obj/<file>.remarks): no GPU needed.

  python tools/kernel_resources.py [out file]          (default profiles/r04_kernel_resources.txt; run `make -C csrc` first)

Columns: VGPRs, AGPRs, spilled VGPRs, scratch bytes per lane, waves per SIMD the registers allow, LDS bytes per workgroup."""
import glob, os, re, subprocess, sys

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "profiles", "r04_kernel_resources.txt")
files = sorted(glob.glob(os.path.join(here, "rte-rrtmgp-cpp_amd", "lib", "obj", "*.remarks")))
if not files:
    sys.exit("no lib/obj/*.remarks: build the device library first (make -C rte-rrtmgp-cpp_amd/csrc)")
ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
ver = next((l.strip() for l in ver.splitlines() if "HIP version" in l), "")
lines = [f"# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, flags of csrc/Makefile; {ver}",
         "%-132s %5s %5s %6s %8s %4s %7s" % ("kernel", "VGPR", "AGPR", "spill", "scratch", "occ", "LDS")]
for f in files:
    rows, cur = {}, None
    for line in open(f, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = rows.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(.+?): (\d+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    names = list(rows)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    lines.append("## " + os.path.basename(f).replace(".remarks", ".hip"))
    table = []
    for n, d in zip(names, dem):
        d = re.sub(r"^void ", "", d); d = d.replace("(anonymous namespace)::", ""); d = re.sub(r"\(.*$", "", d)
        table.append((d, rows[n]))
    for d, r in sorted(table, key=lambda x: x[0]):
        lines.append("%-132s %5d %5d %6d %8d %4d %7d" % (d[:132], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1),
                     r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))
open(out_path, "w").write("\n".join(lines) + "\n")
print(out_path, len(lines), "lines")
