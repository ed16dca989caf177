mkdir -p gpurun_out/r9
timeout -k 10 600 python tools/dev_sizes.py > gpurun_out/r9/sizes.log 2>&1; cat gpurun_out/r9/sizes.log
timeout -k 10 600 python tools/dev_fft2d_sizes.py > gpurun_out/r9/fft2d.log 2>&1; cat gpurun_out/r9/fft2d.log
