#!/bin/bash
out=gpurun_out/ab_lanes11.txt
: > $out
run() {
env $E python bench.py "$@" --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$E $*: %.2f ms/step %.0f Mvoxels/s groups %s' % (d['ms_per_step'], d['value'], d['config'].get('groups_per_gpu')))" >> $out
}
for i in 1 2; do
E="A=1"; run --config 4 --steps 40
E="NDMPS_EARLY_WS_FREE=1"; run --config 4 --steps 40
done
E="A=1"; run --config 2
cat $out
