# run bench.py for a list of --sw-variant / --lw-variant values on the GPU box: bash tools/ab_variants.sh sw 0 11 12 ...
kind=$1; shift
for v in "$@"; do
  timeout -k 10 200 python bench.py --cpu-cols 0 --$kind-variant $v > gpurun_out/abv_${kind}_$v.log 2>&1
  tail -1 gpurun_out/abv_${kind}_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kind variant $v:', d['value'], d['ms_per_step'], 'sw', d['stages']['sw_solver']['ms'], 'lw', d['stages']['lw_solver']['ms'])"
done
