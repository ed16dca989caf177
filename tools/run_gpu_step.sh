mkdir -p gpurun_out/r20
timeout -k 10 300 python tools/dev_ab.py barc4dip_amd/csrc/libb4d_base.so barc4dip_amd/csrc/libb4d.so > gpurun_out/r20/ab.log 2>&1; cat gpurun_out/r20/ab.log
