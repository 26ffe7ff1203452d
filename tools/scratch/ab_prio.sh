#!/bin/bash
out=gpurun_out/ab_turn_prio.txt
: > $out
run() {
env $E python bench.py "$@" --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$E $*: %.2f ms/step %.0f Mvoxels/s fallbacks %s team %.2f ms' % (d['ms_per_step'], d['value'], d['team_fallbacks'], d['roofline']['tridiagonalisation']['launch_us']/1e3))" >> $out
}
NDMPS_TURN_PRIO=1 python -m pytest tests -m gpu -x -q -k "topk_solver or two_halves or concurrent or queues" 2>&1 | tail -1 >> $out
for i in 1 2; do
E="A=1"; run --config metric
E="NDMPS_TURN_PRIO=1"; run --config metric
E="A=1"; run --config 2
E="NDMPS_TURN_PRIO=1"; run --config 2
done
cat $out
