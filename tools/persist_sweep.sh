#!/bin/bash
for spl in 1 20 50; do
  for args in "--steps 20 --warmup 5" "--steps 200 --warmup 50" "--steps 2000 --warmup 200 --streams 1" "--steps 2000 --warmup 200 --streams 2"; do
    echo -n "spl=$spl $args  "
    COMMARL_STEPS_PER_LAUNCH=$spl python bench.py $args --no-train-loop --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value']/1e6, 1), 'M/s', round(j['ms_per_step']*1e3, 2), 'us/step', j['config'].get('streams'))
"
  done
done
