#!/bin/bash
# Collect PMC counters for the bench kernels (one rocprofv3 pass per counter group, no tracing
# domains combined with --pmc other than --kernel-trace).  Usage: tools/pmc_run.sh <outdir> <pmc list...>
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --frames 256 --no-cpu --no-secondary > "$OUT.json" 2> "$OUT.err"
