mkdir -p gpurun_out/r2u
timeout -k 10 900 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_signal.py tests/test_gpu_metrics.py -x -q -m gpu > gpurun_out/r2u/pytest.log 2>&1; tail -5 gpurun_out/r2u/pytest.log
timeout -k 10 300 python tools/dev_general_track.py 2>&1 | tail -2
