# scratch driver for one gpurun call: the GPU tier of the tests, the smoke check, the default bench line and a 2-rank rehearsal of
# the self-spawning bench (gloo: both ranks share the one GPU of the box)
mkdir -p gpurun_out/step
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/step/pytest.log 2>&1; tail -4 gpurun_out/step/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/step/smoke.log 2>&1; tail -1 gpurun_out/step/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/step/bench.json 2> gpurun_out/step/bench.err; echo "bench rc $?"; tail -c 400 gpurun_out/step/bench.json
B4D_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --frames 64 --steps 5 --no-cpu > gpurun_out/step/bench2.json 2> gpurun_out/step/bench2.err; echo "bench2 rc $?"; tail -c 300 gpurun_out/step/bench2.json
