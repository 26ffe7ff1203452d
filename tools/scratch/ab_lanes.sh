#!/bin/bash
out=gpurun_out/ab_lanes12.txt
: > $out
run() {
python bench.py "$@" --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*: %.2f ms/step %.0f Mvoxels/s fallbacks %s' % (d['ms_per_step'], d['value'], d.get('team_fallbacks')))" >> $out
}
for i in 1 2; do for l in 1 2 3; do run --config 5 --lanes $l; done; done
cat $out
