#!/bin/bash
# Five consecutive bench processes on ONE box: `value` on the plan as the library places it (--tune 0, the default) beside the same
# K steps after b4d_plan_tune picked the fastest of six candidate workspaces (plan_tune_comparison).  -> profiles/r03_placement.txt
O=${1:-gpurun_out/placement}
mkdir -p $O
echo "# tools/placement_runs.sh: preheat ${PREHEAT:-default} s"
for i in 1 2 3 4 5; do
  python3 bench.py --no-cpu --no-secondary --steps 20 --warmup 5 --tune-compare 6 ${PREHEAT:+--preheat $PREHEAT} > $O/run$i.json 2> $O/run$i.err
  python3 - $O/run$i.json $i <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = d.get("plan_tune_comparison") or {}
k = d["roofline"]["kernel_ms_per_step"]
print(f"process {sys.argv[2]}: value {d['value']:.0f} frames/s (ms/step {d['ms_per_step']:.3f}; K1 {k['row_r2c']:.3f} K2 {k['col']:.3f} K3 {k['row_c2r']:.3f}) | "
      f"after b4d_plan_tune(6): {c.get('value_with_plan_tune', float('nan')):.0f} frames/s, kept {c.get('kept_ms_per_pass', float('nan')):.3f} / slowest "
      f"{c.get('slowest_ms_per_pass', float('nan')):.3f} ms per pass | ratio value / tuned {d['value'] / c.get('value_with_plan_tune', float('nan')):.4f}", flush=True)
PY
done
