mkdir -p gpurun_out/r9
timeout -k 10 1100 python -m pytest tests/test_gpu_signal.py tests/test_gpu_wiener.py tests/test_gpu_tracking.py -x -q -m gpu --durations=8 > gpurun_out/r9/pytest.log 2>&1; tail -25 gpurun_out/r9/pytest.log
