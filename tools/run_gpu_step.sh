mkdir -p gpurun_out/r10
timeout -k 10 900 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_metrics.py -x -q -m gpu > gpurun_out/r10/pytest.log 2>&1; tail -5 gpurun_out/r10/pytest.log
