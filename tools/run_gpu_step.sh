mkdir -p gpurun_out/r24
timeout -k 10 900 python -m pytest tests/test_gpu_metrics.py -x -q -m gpu > gpurun_out/r24/pytest.log 2>&1; tail -4 gpurun_out/r24/pytest.log
timeout -k 10 300 python tools/dev_stack_prof.py 2>&1 | grep -v amdgpu
