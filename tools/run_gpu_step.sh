mkdir -p gpurun_out/r2z
timeout -k 10 900 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_metrics.py -x -q -m gpu > gpurun_out/r2z/pytest.log 2>&1; tail -3 gpurun_out/r2z/pytest.log
timeout -k 10 300 python tools/bench_configs.py 3 2>&1 | tail -1
bash tools/prof_stats.sh r2z/cfg3b tools/bench_configs.py 3 > /dev/null; python3 tools/prof_summary.py gpurun_out/r2z/cfg3b > gpurun_out/r2z/s.txt; head -4 gpurun_out/r2z/s.txt
