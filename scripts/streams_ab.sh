# bench lines with 1, 2 and 3 contexts (streams) issuing consecutive batches, then the other workloads at the default (2)
for st in 1 2 3; do python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-e2e --streams $st 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('streams',d['streams'],'value %.4g'%d['value'],'ms/step %.4f'%d['ms_per_step'],'single',d['single_stream'],'roof %.3f'%d['roofline']['frac'], d.get('pipelined'))
"; done
for w in config3_per_gpu config4_sample config4_1pct config5_k6; do python3 bench.py --workload $w --steps 300 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$w streams',d['streams'],'value %.4g'%d['value'],'ms/step %.4f'%d['ms_per_step'],'single',d['single_stream'],'roof %.3f'%d['roofline']['frac'], d.get('pipelined'))
"; done
