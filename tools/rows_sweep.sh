#!/bin/bash
# rollout throughput of the headline config against rows per workgroup (COMMARL_FWD_ROWS) and envs per GPU
for rows in 16 32; do
  for envs in 4096 16384; do
    echo -n "rows=$rows envs=$envs  "
    COMMARL_FWD_ROWS=$rows python bench.py --config pp_map10 --envs $envs --steps 1000 --warmup 100 --no-train-loop --no-cpu-baseline --streams 1 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value']/1e6, 1), 'M/s', j['ms_per_step'], {k: v.get('us') for k, v in j['roofline'].get('kernels', {}).items()})
"
  done
done
