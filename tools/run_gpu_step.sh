mkdir -p gpurun_out/r2f
timeout -k 10 900 python -m pytest tests/test_gpu_wiener.py tests/test_gpu_prep.py tests/test_gpu_signal.py -x -q -m gpu > gpurun_out/r2f/pytest.log 2>&1; tail -3 gpurun_out/r2f/pytest.log
for f in 1 4 8; do
B4D_WIENER_FPL=$f bash tools/prof_stats.sh r2f/base$f tools/dev_cfg5.py - 32 > /dev/null
echo "== base FPL $f"; python3 tools/prof_summary.py gpurun_out/r2f/base$f | head -3; tail -1 gpurun_out/r2f/base$f.log
done
for f in 1 2 4 8 16; do B4D_WIENER_FPL=$f python tools/dev_cfg5.py - 32 2>&1 | tail -1; done
