mkdir -p gpurun_out/r2i
timeout -k 10 900 python -m pytest tests/test_gpu_tracking.py -x -q -m gpu -s > gpurun_out/r2i/pytest.log 2>&1; grep -n "observed\|passed\|failed\|Error\|assert" gpurun_out/r2i/pytest.log | head -20
