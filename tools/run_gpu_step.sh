mkdir -p gpurun_out/r8
timeout -k 10 900 python -m pytest tests/test_gpu_signal.py -x -q -m gpu > gpurun_out/r8/pytest.log 2>&1; tail -15 gpurun_out/r8/pytest.log
