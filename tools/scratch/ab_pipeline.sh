#!/bin/bash
out=gpurun_out/ab_pipeline.txt
python -m pytest tests -m gpu -x -q -k "two_halves or from_tensors or concurrent or to_tensors or group_state" 2>&1 | tail -3 > $out
for i in 1 2 3; do
for flag in "--no-pipeline" ""; do
python bench.py --skip-single --no-cpu-baseline --no-configs $flag 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline $flag: %.2f ms/step %.0f Mvoxels/s' % (d['ms_per_step'], d['value']))" >> $out
done
done
for c in 2 4 3; do
for flag in "--no-pipeline" ""; do
python bench.py --config $c --skip-single --no-cpu-baseline --no-configs $flag 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config $c $flag: %.2f ms/step %.0f Mvoxels/s' % (d['ms_per_step'], d['value']))" >> $out
done
done
cat $out
