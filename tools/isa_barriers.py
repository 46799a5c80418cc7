#!/usr/bin/env python3
"""Print the instructions ahead of every s_barrier of one kernel in a device assembly file (hipcc -S --cuda-device-only),
and a count of VALU / LDS / VMEM / scratch instructions:  isa_barriers.py file.s <mangled-name prefix> [context lines]"""
import sys, re
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]; ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 5
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and ':' in l.split()[0])
end = start
while 's_endpgm' not in lines[end]: end += 1
body = lines[start:end]
print(lines[start].split(':')[0], len(body), "lines")
for i, l in enumerate(body):
    if 's_barrier' in l:
        print('---- line', i)
        for k in range(max(0, i-ctx), i+1): print(body[k])
cnt = {}
for l in body:
    t = l.strip().split(' ')[0]
    for key, pat in (("valu", r"^v_"), ("ds", r"^ds_"), ("global_load", r"^global_load"), ("global_store", r"^global_store"),
                     ("scratch", r"^scratch_"), ("s_waitcnt", r"^s_waitcnt"), ("s_barrier", r"^s_barrier")):
        if re.match(pat, t): cnt[key] = cnt.get(key, 0) + 1
print(cnt)
