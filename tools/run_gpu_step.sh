mkdir -p gpurun_out/r24
timeout -k 10 300 python tools/dev_stack_cprof.py speckle > gpurun_out/r24/cprof_speckle2.log 2>&1; grep -v amdgpu gpurun_out/r24/cprof_speckle2.log | head -40 | cut -c1-150
timeout -k 10 300 python tools/dev_stack_cprof.py sharp > gpurun_out/r24/cprof_sharp.log 2>&1; grep -v amdgpu gpurun_out/r24/cprof_sharp.log | head -36 | cut -c1-150
