#!/bin/bash
# A/B on the GPU box: fp32 solver geometries by variant switch (no rebuild): lw/sw = 0/0 (K9 W4, 2 waves/SIMD), 16/10 (K6 W6, 3 waves),
# 17/11 (K5 W8, 4 waves), 15/9 (two columns per lane, rounds 1-3)
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "0 0" "16 10" "17 11" "15 9"; do
  set -- $v
  for a in "" "--allsky --ncol 32768"; do
    timeout -k 10 200 python bench.py --cpu-cols 0 --dtype f32 --lw-variant $1 --sw-variant $2 $a > gpurun_out/ab_v.log 2>&1
    tail -1 gpurun_out/ab_v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lw $1 sw $2 $a', d['ms_per_step'], {k:round(v['ms'],2) for k,v in d['stages'].items() if v['ms']>0.1})" || tail -5 gpurun_out/ab_v.log
  done
done
