mkdir -p gpurun_out/chk2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/chk2/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/chk2/pytest.log
SKIP_TESTS=1 bash tools/gpu_check.sh chk2 "" "--dtype=f32" "--allsky" "--dtype=f32 --allsky --ncol=32768" "--ncol=2048"
