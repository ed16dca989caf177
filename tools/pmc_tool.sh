#!/bin/bash
# PMC counters for an arbitrary python tool.  usage: tools/pmc_tool.sh <outname> <script.py> <script args...> -- <counters...>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
SCRIPT=$1; shift
ARGS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ARGS+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/$SCRIPT" "${ARGS[@]}" > "$OUT.log" 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summarize.py "$OUT" | cut -c1-600
