mkdir -p gpurun_out/r23
timeout -k 10 600 python tools/dev_soak_routes.py 5 150 > gpurun_out/r23/soak_routes.log 2>&1; tail -6 gpurun_out/r23/soak_routes.log
