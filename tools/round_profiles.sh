#!/bin/bash
# Round-end measurement pass on ONE GPU box: every profiles/rNN_* file regenerated from the library as it
# ships.  usage: bash tools/round_profiles.sh r03   -> gpurun_out/<tag>/..., copied to profiles/ by the builder afterwards.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
# 1. the driver's line (default invocation) and the same command under rocprofv3 --kernel-trace --stats
python3 bench.py > $O/bench.json 2> $O/bench.err < /dev/null
tail -c 300 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu > $O/bench_under_rocprof.json 2> $O/stats.err < /dev/null
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
# 2. HBM traffic of the headline kernels (FETCH_SIZE / WRITE_SIZE in passes of their own)
cd $R
bash tools/pmc_run.sh $O/pmc_fetch FETCH_SIZE < /dev/null
bash tools/pmc_run.sh $O/pmc_write WRITE_SIZE < /dev/null
python3 tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write 256 $O/pmc_col.json > $O/pmc_col.log 2>&1
# 3. the secondary legs on their own: kernel stats + traffic counters (fft2d, cfg3, cfg5)
for leg in fft2d cfg3 cfg5; do
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${leg}_stats -- python3 $R/tools/dev_secondary.py $leg > $O/${leg}.log 2>&1 < /dev/null
  f=$(find $O/${leg}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${leg}_kernel_stats.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${leg}_fetch -- python3 $R/tools/dev_secondary.py $leg > /dev/null 2>&1 < /dev/null
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${leg}_write -- python3 $R/tools/dev_secondary.py $leg > /dev/null 2>&1 < /dev/null
  cd $R
  { echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/dev_secondary.py $leg; per-launch averages in KB as"; \
    echo "# rocprofv3 reports them (FETCH_SIZE is half of the bytes of a wide streamed read on gfx950: MI355X_MICROARCH.md)"; \
    python3 tools/pmc_summarize.py $O/${leg}_fetch $O/${leg}_write; } > $O/pmc_${leg}.txt 2>&1
  tail -2 $O/${leg}.log | cut -c1-300
done
# 4. sizes beside the headline (pipeline and fft2d)
python3 tools/dev_sizes.py > $O/sizes.txt 2>&1
python3 tools/dev_fft2d_pow2.py >> $O/sizes.txt 2>&1
python3 tools/bench_aggregators.py > $O/aggregators.jsonl 2>&1
head -8 $O/bench_kernel_stats.csv | cut -c1-150
cat $O/pmc_col.log | head -40
