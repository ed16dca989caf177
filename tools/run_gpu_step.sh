mkdir -p gpurun_out/r7
B4D_LIB=barc4dip_amd/csrc/libb4d_ALIGNED_MIRROR.so timeout -k 10 300 python tools/dev_colseg.py 2048 > gpurun_out/r7/colseg_am.log 2>&1; cat gpurun_out/r7/colseg_am.log
