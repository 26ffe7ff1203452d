#!/bin/bash
out=gpurun_out/ab_groups.txt
: > $out
for c in metric 2 4; do
for g in 1 2 3 4; do
python bench.py --config $c --groups $g --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config $c groups $g: %.2f ms/step %.0f Mvoxels/s' % (d['ms_per_step'], d['value']))" >> $out
done
done
cat $out
