#!/bin/bash
# Fixed cost vs per-step cost of the timed region: bench.py at several --steps / --chunk / stream settings.
# usage (GPU box): bash tools/steps_sweep.sh > gpurun_out/steps_sweep.txt
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --streams 1" "--steps 20 --warmup 5 --streams 2" "--steps 20 --warmup 5 --streams 1 --chunk 5" "--steps 40 --warmup 5 --streams 1" "--steps 80 --warmup 5 --streams 1" \
            "--steps 200 --warmup 50 --streams 1" "--steps 200 --warmup 50 --streams 2" "--steps 2000 --warmup 200 --streams 1" "--steps 2000 --warmup 200 --streams 2" "--steps 2000 --warmup 200 --streams 2 --chunk 200" "--steps 2000 --warmup 200 --streams 4"; do
  python bench.py $args --no-train-loop --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args', '->', round(d['value']/1e6,1), 'M  total_ms', round(d['ms_per_step']*d['steps'],4), ' ms/step', round(d['ms_per_step'],5), d['config']['streams'], d['config']['graphs']['chunk_lengths'])
"
done
