mkdir -p gpurun_out/r5
timeout -k 10 300 python tools/dev_diag_pa.py 1 > gpurun_out/r5/diag.log 2>&1; timeout -k 10 300 python tools/dev_diag_pa.py 2 >> gpurun_out/r5/diag.log 2>&1; timeout -k 10 300 python tools/dev_diag.py >> gpurun_out/r5/diag.log 2>&1; cat gpurun_out/r5/diag.log
