#!/usr/bin/env python3
"""Instruction mix of the main loop (the longest backward-branch span) of one kernel in a device assembly file
(hipcc -S --cuda-device-only):   isa_mix.py file.s <mangled-name prefix>"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and ':' in l.split()[0])
end = start
while 's_endpgm' not in lines[end]: end += 1
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
best = (0, 0, 0)
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
        best = (i - labels[m.group(1)], labels[m.group(1)], i)
loop = body[best[1]:best[2]]
c = collections.Counter()
for l in loop:
    t = l.strip().split(' ')[0]
    if re.match(r'^(v_|s_|ds_|global_|scratch_|buffer_)', t): c[t] += 1
groups = collections.Counter()
for k, v in c.items():
    if k.startswith('v_pk'): groups['v_pk_*'] += v
    elif re.match(r'v_(fma|fmac|fmamk|fmaak|mul|add|sub|mac|mad)_f', k): groups['fma/mul/add'] += v
    elif re.match(r'v_(exp|rcp|rsq|sqrt|log)', k): groups['transcendental'] += v
    elif re.match(r'v_(max|min|med)', k): groups['max/min'] += v
    elif k.startswith('v_cndmask'): groups['cndmask'] += v
    elif k.startswith('v_cmp'): groups['cmp'] += v
    elif re.match(r'v_mov|v_accvgpr', k): groups['mov'] += v
    elif k.startswith('v_'): groups['valu ' + k] += v
    elif k.startswith('ds_'): groups['ds'] += v
    elif k.startswith('s_'): groups['salu'] += v
    else: groups[k] += v
print(lines[start].split(':')[0][-60:], '| loop lines', len(loop), '| VALU', sum(v for k, v in c.items() if k.startswith('v_')))
for k, v in groups.most_common(24): print('   ', k, v)
