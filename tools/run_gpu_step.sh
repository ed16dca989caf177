mkdir -p gpurun_out/r11
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r11/pytest.log 2>&1; tail -4 gpurun_out/r11/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r11/smoke.log 2>&1; tail -2 gpurun_out/r11/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/r11/bench.json 2> gpurun_out/r11/bench.err; echo "bench rc $?"; cat gpurun_out/r11/bench.json
