#!/bin/bash
# Fixed cost vs per-step cost of the timed region: bench.py at several --steps / stream / fused-in-graph settings.
# usage (GPU box): bash tools/steps_sweep.sh [config] > gpurun_out/steps_sweep.txt
CFG=${1:-pp_map10}
for fused in 0 1; do
for args in "--steps 20 --warmup 5 --streams 1" "--steps 20 --warmup 5 --streams 2" "--steps 200 --warmup 50 --streams 1" "--steps 2000 --warmup 200 --streams 1" "--steps 2000 --warmup 200 --streams 2"; do
  COMMARL_GRAPH_FUSED=$fused python bench.py --config $CFG $args --no-train-loop --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fused_in_graph=$fused $args', '->', round(d['value']/1e6,2), 'M  total_ms', round(d['ms_per_step']*d['steps'],4), ' us/step', round(d['ms_per_step']*1e3,2), d['config']['streams'], d['config']['graphs']['chunk_lengths'])
"
done; done
