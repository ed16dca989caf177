mkdir -p gpurun_out/r2h
timeout -k 10 900 python -m pytest tests/test_gpu_stats.py tests/test_gpu_large.py -x -q -m gpu > gpurun_out/r2h/pytest.log 2>&1; tail -5 gpurun_out/r2h/pytest.log
timeout -k 10 600 python bench.py > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; echo "bench rc $?"; tail -3 gpurun_out/r2h/bench.err
python - <<'PY'
import json
l=json.load(open("gpurun_out/r2h/bench.json"))
print(l["value"], l["roofline"]["frac"], l["pipeline_roofline"]["frac"])
print(json.dumps(l["secondary"], indent=1))
print(l["cpu_baseline"]["value"])
PY
