#!/usr/bin/env python3
"""Where a kernel waits on the vector-memory counter: prints every `s_waitcnt vmcnt(N)` of one kernel in a device assembly file
(hipcc -S --cuda-device-only) with the basic block it sits in and the VMEM instructions of that block, marking blocks that are loops
(a branch back to their own label):   isa_waits.py file.s <mangled-name prefix>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and ':' in l.split()[0])
end = start
while 's_endpgm' not in lines[end]: end += 1
body = lines[start:end]
blocks = []; cur = ["<entry>", []]
for l in body:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
    else:
        cur[1].append(l.strip())
blocks.append(cur)
for name, ins in blocks:
    vm = [i for i in ins if re.match(r'^(global_|buffer_|scratch_|flat_)', i)]
    waits = [i for i in ins if i.startswith('s_waitcnt') and 'vmcnt' in i]
    loop = any(re.match(r'^s_cbranch\w*\s+' + re.escape(name) + r'\b', i) or re.match(r'^s_branch\s+' + re.escape(name) + r'\b', i) for i in ins)
    if not waits: continue
    nst = sum(1 for i in vm if 'store' in i); nld = len(vm) - nst
    print(f"{name:14s} {'LOOP' if loop else '    '} instrs {len(ins):4d} loads {nld:3d} stores {nst:3d}  waits: " + ", ".join(w.replace('s_waitcnt ', '') for w in waits))
if len(sys.argv) > 3 and sys.argv[3] == "stores":
    print("---- blocks with stores: VMEM instructions and vmcnt waits in order")
    for name, ins in blocks:
        if not any(re.match(r'^(global|buffer|flat)_store', i) for i in ins): continue
        seq = []
        for i in ins:
            if re.match(r'^(global_|buffer_|scratch_|flat_)', i): seq.append(i.split()[0] + (" nt" if " nt" in i else ""))
            elif i.startswith('s_waitcnt') and 'vmcnt' in i: seq.append("[" + i.replace('s_waitcnt ', '') + "]")
            elif i.startswith('s_barrier'): seq.append("[barrier]")
            elif re.match(r'^s_c?branch', i): seq.append("->" + i.split()[-1])
        print(name, " ".join(seq))
