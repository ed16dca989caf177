mkdir -p gpurun_out/r2q
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2q/pytest.log 2>&1; tail -5 gpurun_out/r2q/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/r2q/bench.json 2> gpurun_out/r2q/bench.err; echo "bench rc $?"
