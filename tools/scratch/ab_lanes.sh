#!/bin/bash
out=gpurun_out/ab_lanes2.txt
: > $out
run() {
python bench.py "$@" --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*: %.2f ms/step %.0f Mvoxels/s fallbacks %s' % (d['ms_per_step'], d['value'], d.get('team_fallbacks')))" >> $out
}
for b in 16 24 31; do for l in 1 2 3; do run --config metric --batch $b --lanes $l --steps 20; done; done
for l in 1 2; do run --config metric --batch 32 --lanes $l --steps 20; done
for b in 16 31 32; do for l in 1 2 3; do run --config 2 --batch $b --lanes $l --steps 20; done; done
cat $out
