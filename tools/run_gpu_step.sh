mkdir -p gpurun_out/r3a
bash tools/prof_stats.sh r3a/cfg3 tools/bench_configs.py 3 > /dev/null; python3 tools/prof_summary.py gpurun_out/r3a/cfg3 > gpurun_out/r3a/cfg3_summary.txt; head -8 gpurun_out/r3a/cfg3_summary.txt
bash tools/pmc_tool.sh r3a/cfg3_fetch tools/bench_configs.py 3 -- FETCH_SIZE | cut -c1-200
bash tools/pmc_tool.sh r3a/cfg3_write tools/bench_configs.py 3 -- WRITE_SIZE | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3a/bench_full -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/r3a/bench_full.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3a/bench_full.err
cd $GRAFT_REPO_ROOT; python3 tools/prof_summary.py gpurun_out/r3a/bench_full | head -20
