mkdir -p gpurun_out/r2p
timeout -k 10 900 python -m pytest tests/test_gpu_signal.py tests/test_gpu_wiener.py -x -q -m gpu > gpurun_out/r2p/pytest.log 2>&1; tail -3 gpurun_out/r2p/pytest.log
for f in 8 16; do B4D_WIENER_FPL=$f python tools/dev_cfg5.py - 32 2>&1 | tail -1; done
B4D_WIENER_FPL=8 bash tools/prof_stats.sh r2p/cfg5 tools/dev_cfg5.py - 32 > /dev/null; python3 tools/prof_summary.py gpurun_out/r2p/cfg5 > gpurun_out/r2p/s.txt; head -3 gpurun_out/r2p/s.txt
timeout -k 10 300 python tools/dev_sizes.py 2>&1 | tail -2
