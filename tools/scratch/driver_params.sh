#!/bin/bash
for sw in "1 0" "2 1" "5 1" "20 3" "3 5"; do
set -- $sw
python bench.py --gpus 1 --steps $1 --warmup $2 --no-cpu-baseline --no-configs --skip-single 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps $1 warmup $2: %.2f ms/step %.0f Mvoxels/s (steps %d warmup %d)' % (d['ms_per_step'], d['value'], d['steps'], d['warmup']))"
done
