#!/bin/bash
# persistent rollout launches inside the captured chunk: steps per launch (COMMARL_STEPS_PER_LAUNCH) x start stagger of the
# second half of the grid (COMMARL_CHUNK_STAGGER, units of ~2048 clocks)
for spl in ${SPLS:-10 50}; do
  for stg in ${STGS:-0 4 8 12 16 20}; do
    for args in "--steps 2000 --warmup 200 --streams 1"; do
      echo -n "spl=$spl stagger=$stg $args  "
      COMMARL_STEPS_PER_LAUNCH=$spl COMMARL_CHUNK_STAGGER=$stg python bench.py $args --no-train-loop --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value']/1e6, 1), 'M/s', round(j['ms_per_step']*1e3, 2), 'us/step')
"
    done
  done
done
