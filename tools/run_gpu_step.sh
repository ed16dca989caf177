mkdir -p gpurun_out/r11
bash tools/pmc_tool.sh r11/cfg3_fetch tools/bench_configs.py 3 -- FETCH_SIZE > gpurun_out/r11/pmc_fetch.txt 2>&1; tail -14 gpurun_out/r11/pmc_fetch.txt
bash tools/pmc_tool.sh r11/cfg3_write tools/bench_configs.py 3 -- WRITE_SIZE > gpurun_out/r11/pmc_write.txt 2>&1; tail -14 gpurun_out/r11/pmc_write.txt
