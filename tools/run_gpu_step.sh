mkdir -p gpurun_out/r18
timeout -k 10 900 python -m pytest tests/test_gpu_metrics.py -x -q -m gpu > gpurun_out/r18/pytest.log 2>&1; tail -25 gpurun_out/r18/pytest.log
