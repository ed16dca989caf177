mkdir -p gpurun_out/r2y
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2y/pytest.log 2>&1; tail -3 gpurun_out/r2y/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2y/bench.json 2> gpurun_out/r2y/bench.err; echo "bench rc $?"
