mkdir -p gpurun_out/r13
timeout -k 10 300 python tools/dev_ab.py barc4dip_amd/csrc/libb4d_base.so barc4dip_amd/csrc/libb4d.so > gpurun_out/r13/ab.log 2>&1; cat gpurun_out/r13/ab.log
timeout -k 10 900 python -m pytest tests/test_gpu_signal.py tests/test_gpu_tracking.py -x -q -m gpu > gpurun_out/r13/pytest.log 2>&1; tail -3 gpurun_out/r13/pytest.log
