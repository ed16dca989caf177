#!/bin/bash
# Round-end measurement pass on the GPU box: headline bench, rocprofv3 kernel stats of the same command, HBM traffic
# counters in their own passes.  Everything lands in gpurun_out/final/.
O=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 bench.py > $O/bench.json 2> $O/bench.err < /dev/null
tail -c 600 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $O/stats.json 2> $O/stats.err < /dev/null
cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh $O/pmc_fetch FETCH_SIZE < /dev/null
bash tools/pmc_run.sh $O/pmc_write WRITE_SIZE < /dev/null
python3 tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write 256 $O/pmc_col.json > $O/pmc_col.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $O/kernel_stats.csv; head -8 $f | cut -c1-160; fi
cat $O/pmc_col.log | head -30
