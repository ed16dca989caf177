mkdir -p gpurun_out/r2s
timeout -k 10 600 python bench.py --no-cpu > gpurun_out/r2s/bench.json 2> gpurun_out/r2s/bench.err; echo "bench rc $?"
python - <<'PY'
import json
l=json.load(open("gpurun_out/r2s/bench.json"))
print(l["value"]); print(json.dumps(l["secondary"]["cfg4"]))
PY
timeout -k 10 300 python tools/dev_general_track.py 2>&1 | tail -3
