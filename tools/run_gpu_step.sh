mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_metrics.py -x -q -m gpu -k "detector_format or default_tracker" > gpurun_out/r3b/pytest.log 2>&1; tail -12 gpurun_out/r3b/pytest.log
