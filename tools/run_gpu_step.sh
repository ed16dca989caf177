mkdir -p gpurun_out/r16
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r16/pytest.log 2>&1; tail -4 gpurun_out/r16/pytest.log
