#!/bin/bash
out=gpurun_out/ab_streamed.txt
: > $out
for n in 1 2 4 8; do
for e in "A=1" "NDMPS_TRD_TEAM_NARROW=1"; do
echo "== $e" >> $out
env $e python tools/scratch/small_batch_stream.py $n 256 64 2>&1 | grep "lanes 3\|sync" | head -2 >> $out
done
done
for e in "A=1" "NDMPS_TRD_TEAM_NARROW=1"; do
env $e python bench.py --config 3 --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e config 3: %.2f ms/step %.0f Mvoxels/s' % (d['ms_per_step'], d['value']))" >> $out
done
cat $out
