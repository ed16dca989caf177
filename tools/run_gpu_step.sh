mkdir -p gpurun_out/r2m
timeout -k 10 900 python -m pytest tests/test_gpu_signal.py tests/test_gpu_wiener.py -x -q -m gpu > gpurun_out/r2m/pytest.log 2>&1; tail -15 gpurun_out/r2m/pytest.log
timeout -k 10 300 python tools/dev_sizes.py 2>&1 | tail -12
