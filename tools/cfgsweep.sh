for c in co_map20 pp_map30 co_map30; do for f in 0 1; do for st in 1 2; do
COMMARL_GRAPH_FUSED=$f python bench.py --config $c --steps 500 --warmup 100 --streams $st --no-train-loop --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c fused=$f streams=$st', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v['us'],1) for k,v in d['roofline']['kernels'].items()})"
done; done; done
