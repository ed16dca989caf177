mkdir -p gpurun_out/r26
timeout -k 10 900 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_metrics.py -x -q -m gpu > gpurun_out/r26/pytest.log 2>&1; tail -5 gpurun_out/r26/pytest.log
bash tools/prof_stats.sh r26/prof_gtrack tools/dev_general_track.py > /dev/null; python3 - <<'PY'
import csv,glob
f=sorted(glob.glob("gpurun_out/r26/prof_gtrack/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:9]:
    print(r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg", r["Percentage"])
PY
