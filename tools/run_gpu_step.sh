mkdir -p gpurun_out/r2g
timeout -k 10 900 python -m pytest tests/test_gpu_wiener.py -x -q -m gpu > gpurun_out/r2g/pytest.log 2>&1; tail -3 gpurun_out/r2g/pytest.log
for lib in - $GRAFT_REPO_ROOT/barc4dip_amd/csrc/libb4d_nopf.so; do
B4D_WIENER_FPL=8 bash tools/prof_stats.sh r2g/p tools/dev_cfg5.py $lib 32 > /dev/null
echo "== $lib FPL 8"; python3 tools/prof_summary.py gpurun_out/r2g/p | head -3
for f in 4 8 16; do B4D_WIENER_FPL=$f python tools/dev_cfg5.py $lib 32 2>&1 | tail -1; done
done
