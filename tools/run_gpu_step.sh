mkdir -p gpurun_out/r12
timeout -k 10 900 python -m pytest tests/test_gpu_tracking.py -x -q -m gpu -k "power_of_two_sizes" > gpurun_out/r12/pytest.log 2>&1; tail -30 gpurun_out/r12/pytest.log
