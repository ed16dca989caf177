mkdir -p gpurun_out/r2l
timeout -k 10 900 python -m pytest tests/test_gpu_wiener.py -x -q -m gpu > gpurun_out/r2l/pytest.log 2>&1; tail -2 gpurun_out/r2l/pytest.log
for f in 8 16; do B4D_WIENER_FPL=$f python tools/dev_cfg5.py - 32 2>&1 | tail -1; done
B4D_WIENER_FPL=8 bash tools/prof_stats.sh r2l/cfg5 tools/dev_cfg5.py - 32 > /dev/null; python3 tools/prof_summary.py gpurun_out/r2l/cfg5 > gpurun_out/r2l/cfg5_summary.txt; head -3 gpurun_out/r2l/cfg5_summary.txt
B4D_WIENER_FPL=8 bash tools/pmc_tool.sh r2l/cfg5_fetch tools/dev_cfg5.py - 16 -- FETCH_SIZE | grep wmr
B4D_WIENER_FPL=8 bash tools/pmc_tool.sh r2l/cfg5_write tools/dev_cfg5.py - 16 -- WRITE_SIZE | grep wmr
