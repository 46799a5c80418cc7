#!/bin/bash
# Round-4 evidence, run ON the GPU box from the repo root: bash tools/profile_r04.sh
# fp64 C4 in both flux modes (tools/profile_round.sh), then BASELINE C5's per-GPU share (fp32 all-sky, 32 768 columns), the strong-
# scaling share of C4 (2 048 columns), fp64 all-sky and fp32 clear sky, each with rocprofv3 kernel statistics and the three PMC passes.
set -o pipefail
bash tools/profile_round.sh r04 > gpurun_out/r04_round.log 2>&1; tail -3 gpurun_out/r04_round.log
bash tools/profile_config.sh r04 f32_allsky_ncol32768 "f32|broadband-allsky|32768x140x256" --dtype f32 --allsky --ncol 32768 > gpurun_out/r04_c5.log 2>&1; tail -2 gpurun_out/r04_c5.log
bash tools/profile_config.sh r04 f64_broadband_ncol2048 "f64|broadband|2048x140x256" --ncol 2048 > gpurun_out/r04_2048.log 2>&1; tail -2 gpurun_out/r04_2048.log
bash tools/profile_config.sh r04 f64_allsky "f64|broadband-allsky|16384x140x256" --allsky > gpurun_out/r04_allsky.log 2>&1; tail -2 gpurun_out/r04_allsky.log
bash tools/profile_config.sh r04 f32_broadband "f32|broadband|16384x140x256" --dtype f32 > gpurun_out/r04_f32.log 2>&1; tail -2 gpurun_out/r04_f32.log
for n in 4096 8192; do timeout -k 10 200 python3 bench.py --cpu-cols 0 --ncol $n | tail -1 > gpurun_out/r04/r04_bench_f64_broadband_ncol$n.json; done
python3 bench.py --cpu-cols 0 --driver cxx | tail -1 > gpurun_out/r04/r04_bench_f64_broadband_driver_cxx.json
python3 tools/driver_c4.py 16384 --timings --async > gpurun_out/r04/r04_driver_c4_timings.txt 2>&1
ls gpurun_out/r04 | head -80
