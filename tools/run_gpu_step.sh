mkdir -p gpurun_out/r4
timeout -k 10 600 python tools/dev_fft2d_sizes.py > gpurun_out/r4/fft2d.log 2>&1; cat gpurun_out/r4/fft2d.log
timeout -k 10 900 python -m pytest tests/test_gpu_signal.py -x -q -m gpu > gpurun_out/r4/pytest.log 2>&1; tail -12 gpurun_out/r4/pytest.log
