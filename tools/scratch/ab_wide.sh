#!/bin/bash
out=gpurun_out/ab_half_launch.txt
: > $out
run() {
env $E python bench.py "$@" --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$E $*: %.2f ms/step %.0f Mvoxels/s fallbacks %s' % (d['ms_per_step'], d['value'], d['team_fallbacks']))" >> $out
}
for i in 1 2; do
E="A=1"; run --config metric
E="NDMPS_TRD_TEAM_HALF=1"; run --config metric
done
E="NDMPS_TRD_TEAM_HALF=1"; run --config metric --lanes 4
E="NDMPS_TRD_TEAM_HALF=1"; run --config metric --groups 2 --lanes 2
cat $out
