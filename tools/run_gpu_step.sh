mkdir -p gpurun_out/r2r
timeout -k 10 600 python -m pytest tests/test_gpu_stats.py tests/test_gpu_wiener.py -x -q -m gpu > gpurun_out/r2r/pytest.log 2>&1; tail -2 gpurun_out/r2r/pytest.log
bash tools/pmc_run.sh $GRAFT_REPO_ROOT/gpurun_out/r2r/pmc_fetch FETCH_SIZE
bash tools/pmc_run.sh $GRAFT_REPO_ROOT/gpurun_out/r2r/pmc_write WRITE_SIZE
python3 tools/pmc_to_json.py gpurun_out/r2r/pmc_fetch gpurun_out/r2r/pmc_write 64 gpurun_out/r2r/r02_pmc_col.json | tail -30
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2r/bench_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r2r/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2r/bench_prof.err
cd $GRAFT_REPO_ROOT; python3 tools/prof_summary.py gpurun_out/r2r/bench_prof > gpurun_out/r2r/bench_summary.txt; cat gpurun_out/r2r/bench_summary.txt
