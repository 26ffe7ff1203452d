#!/bin/bash
out=gpurun_out/ab_oneturn.txt
: > $out
run() {
env $E python bench.py "$@" --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$E $*: %.2f ms/step %.0f Mvoxels/s gram frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))" >> $out
}
for i in 1 2; do
for c in metric 2 3; do
E="A=1"; run --config $c
E="NDMPS_ONE_TURN=1"; run --config $c
done
done
cat $out
