mkdir -p gpurun_out/r14
timeout -k 10 900 python -m pytest tests/test_gpu_wiener.py -x -q -m gpu -k "unsupervised or uw_step" > gpurun_out/r14/pytest.log 2>&1; tail -40 gpurun_out/r14/pytest.log
