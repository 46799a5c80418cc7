#!/bin/bash
# Round-4 side tables, run ON the GPU box after tools/profile_r04.sh: other shapes (odd column counts, tall columns), column spread
# with and without sorting, the instruction-issue and store-pattern probes. Writes gpurun_out/r04x/.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04x; mkdir -p $OUT
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms  ', round(d['value']), d['unit'], ' hand-backs', (d.get('gas_window') or {}).get('frac'), ' sorted', d['config'].get('columns_sorted'), {k:round(v['ms'],2) for k,v in d['stages'].items() if v['ms']>0.02})"; }
{
echo "# bench.py --cpu-cols 0 <arguments>: ms per LW+SW solve, columns/s, stage times"
for a in "--ncol 16384" "--ncol 16385" "--ncol 16000" "--ncol 16001" "--ncol 4096 --nlay 287" "--ncol 4096 --nlay 288" "--ncol 4096 --nlay 400" "--ncol 4096 --nlay 512" \
         "--ngpt 128 --nbnd 16" "--ngpt 224 --nbnd 14" "--dtype f32 --ncol 32768" "--dtype f32 --allsky --ncol 32768"; do
  timeout -k 10 300 python3 bench.py --cpu-cols 0 --steps 10 $a 2>/dev/null | line "[$a]"
done
echo "# hand-back census of the 224 / 14 set (RRX_GW_STATS=1):"
RRX_GW_STATS=1 timeout -k 10 300 python3 bench.py --cpu-cols 0 --steps 1 --warmup 1 --ngpt 224 --nbnd 14 2> $OUT/census.err > /dev/null; grep -h "handed back" $OUT/census.err | sort | uniq -c | head -6
} > $OUT/r04_other_shapes.txt
{
for s in 0 0.05 0.35; do for so in 0 auto; do
  timeout -k 10 200 python3 bench.py --cpu-cols 0 --steps 10 --col-spread $s --sort-columns $so 2>/dev/null | line "col-spread $s sort-columns $so:"
done; done
} > $OUT/r04_col_spread.txt
mkdir -p tools/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_build/issue_mix_bench tools/issue_mix_bench.hip > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/_build/store_pattern_bench tools/store_pattern_bench.hip > /dev/null 2>&1
{ echo "# tools/issue_mix_bench.hip: cycles per instruction per SIMD at 1..5 resident waves per SIMD"; timeout -k 10 120 tools/_build/issue_mix_bench; } > $OUT/r04_issue_mix.txt 2>&1
{ echo "# tools/store_pattern_bench.hip: the windowed gas optics' write pattern by itself (16 384 x 140 x 256)"; timeout -k 10 120 tools/_build/store_pattern_bench; } > $OUT/r04_store_pattern.txt 2>&1
tail -3 $OUT/r04_other_shapes.txt; tail -2 $OUT/r04_col_spread.txt
