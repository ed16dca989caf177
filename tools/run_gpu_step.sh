mkdir -p gpurun_out/r22
timeout -k 10 900 python -m pytest tests/test_gpu_metrics.py -x -q -m gpu > gpurun_out/r22/pytest.log 2>&1; tail -4 gpurun_out/r22/pytest.log
timeout -k 10 300 python tools/dev_aggr_prof.py 2>&1 | grep -v amdgpu
bash tools/prof_stats.sh r22/prof_aggr2 tools/dev_aggr_prof.py > /dev/null; python3 - <<'PY'
import csv,glob
f=sorted(glob.glob("gpurun_out/r22/prof_aggr2/*/*kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f)))
for r in rows[:6]:
    print(r["Name"][:72].ljust(72), r["Calls"], round(int(r["TotalDurationNs"])/1e3), "us", r["Percentage"])
PY
