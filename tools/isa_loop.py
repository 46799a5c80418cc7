#!/usr/bin/env python3
"""Instruction histogram of the innermost loop that holds non-temporal stores (the g-point loop of the windowed gas optics) of one kernel
in a device assembly file:   isa_loop.py file.s <mangled-name prefix> [which loop, 0 = innermost]"""
import re, collections, sys
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]; which = int(sys.argv[3]) if len(sys.argv) > 3 else 0
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and ':' in l.split()[0])
end = start
while 's_endpgm' not in lines[end]: end += 1
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i: loops.append((labels[m.group(1)], i))
cand = sorted([(a, b) for a, b in loops if any('_store_' in x and ' nt' in x for x in body[a:b])], key=lambda ab: ab[1] - ab[0])
print(lines[start].split(':')[0][-50:], '| loops with nt stores (lines):', [b - a for a, b in cand][:6])
a, b = cand[which]
c = collections.Counter()
for l in body[a:b]:
    t = l.strip().split(' ')[0]
    if re.match(r'^(v_|s_|ds_|global_|scratch_|buffer_)', t): c[t] += 1
print('  VALU', sum(v for k, v in c.items() if k.startswith('v_')), '| ds', sum(v for k, v in c.items() if k.startswith('ds_')),
      '| salu', sum(v for k, v in c.items() if k.startswith('s_')), '| vmem', sum(v for k, v in c.items() if k.startswith(('global_', 'scratch_', 'buffer_'))))
print('  ', [(k, v) for k, v in c.most_common(40) if k.startswith(('v_', 'ds_', 'global', 'scratch'))])
