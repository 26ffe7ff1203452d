#!/bin/bash
for c in 4 metric 3; do
python bench.py --config $c --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('config $c: %.2f ms/step; reshape_stage' % d['ms_per_step'], {k: round(v,3) for k,v in r['reshape_stage'].items() if k!='kernel'}, 'group', {k: (round(v,3) if isinstance(v,float) else v) for k,v in r.get('reshape_stage_group',{}).items() if k!='kernel'})"
done
