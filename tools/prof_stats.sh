#!/bin/bash
# rocprofv3 kernel-trace + stats of one python tool; prints the head of the kernel stats CSV.
# usage: tools/prof_stats.sh <outname> <script.py> [args...]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$OUT.log" 2>&1 < /dev/null
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -14 "$f" | cut -c1-220; else echo "no stats file"; tail -5 "$OUT.log"; fi
