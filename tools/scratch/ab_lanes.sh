#!/bin/bash
out=gpurun_out/ab_oneturn_lanes.txt
: > $out
run() {
env $E python bench.py "$@" --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$E $*: %.2f ms/step %.0f Mvoxels/s gram frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))" >> $out
}
for l in 3 4 5 6; do
E="NDMPS_ONE_TURN=1"; run --config metric --groups 1 --lanes $l --steps 20
done
E="A=1"; run --config metric --groups 1 --lanes 5 --steps 20
E="NDMPS_ONE_TURN=1"; run --config metric --groups 2 --lanes 2 --steps 20
cat $out
