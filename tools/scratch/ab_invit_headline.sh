#!/bin/bash
out=gpurun_out/ab_invit_headline.txt
: > $out
for cb in 128 32 16 128 32 16; do
  NDMPS_INVIT_CB=$cb python bench.py --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline NDMPS_INVIT_CB=$cb: %.2f ms/step %.0f Mvoxels/s' % (d['ms_per_step'], d['value']))" >> $out
done
for cb in 128 32 16; do
  echo "== trd_probe NDMPS_INVIT_CB=$cb" >> $out
  NDMPS_INVIT_CB=$cb python tools/trd_probe.py 32 512 64 2>&1 | grep -v amdgpu | tail -4 >> $out
  NDMPS_INVIT_CB=$cb python tools/trd_probe.py 1 512 64 2>&1 | grep -v amdgpu | tail -4 >> $out
done
cat $out
