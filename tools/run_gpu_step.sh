mkdir -p gpurun_out/r12
cp barc4dip_amd/csrc/libb4d_ns.so barc4dip_amd/csrc/libb4d.so
bash tools/prof_stats.sh r12/prof_cfg3_nosel tools/bench_configs.py 3 | awk -F'","' '{print substr($1,1,70), $2, $3, $4, $5}' | head -8
