mkdir -p gpurun_out/r19
timeout -k 10 300 python tools/dev_ab.py barc4dip_amd/csrc/libb4d.so barc4dip_amd/csrc/libb4d_tw1.so > gpurun_out/r19/ab1.log 2>&1; cat gpurun_out/r19/ab1.log
timeout -k 10 300 python tools/dev_ab.py barc4dip_amd/csrc/libb4d.so barc4dip_amd/csrc/libb4d_tw2.so > gpurun_out/r19/ab2.log 2>&1; cat gpurun_out/r19/ab2.log
