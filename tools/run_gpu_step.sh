mkdir -p gpurun_out/r2x
timeout -k 10 900 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_metrics.py tests/test_gpu_signal.py -x -q -m gpu > gpurun_out/r2x/pytest.log 2>&1; tail -4 gpurun_out/r2x/pytest.log
timeout -k 10 300 python tools/bench_configs.py 3 2>&1 | tail -1
bash tools/prof_stats.sh r2x/cfg3 tools/bench_configs.py 3 > /dev/null; python3 tools/prof_summary.py gpurun_out/r2x/cfg3 > gpurun_out/r2x/s.txt; head -7 gpurun_out/r2x/s.txt
timeout -k 10 200 python tools/dev_soak_track.py 11 2>&1 | tail -1
