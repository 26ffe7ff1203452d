#!/bin/bash
out=gpurun_out/ab_invit2.txt
: > $out
for cb in 128 32 16; do
echo "== NDMPS_INVIT_CB=$cb" >> $out
NDMPS_INVIT_CB=$cb python tools/trd_probe.py 32 512 64 2>&1 | grep "invit phases\|B=" | cut -c1-150 >> $out
NDMPS_INVIT_CB=$cb python tools/trd_probe.py 1 512 64 2>&1 | grep "invit phases\|B=" | cut -c1-150 >> $out
done
python tools/panel_probe.py 1024,2048 128 1 2>&1 | grep panel >> $out
python tools/full_probe.py 2048,4096 --no-jacobi 2>&1 | grep -v amdgpu >> $out
python -m pytest tests -m gpu -x -q -k "topk or eigensolver or lapack or solver or exact or f64" 2>&1 | tail -2 >> $out
for i in 1 2; do
python bench.py --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline: %.2f ms/step %.0f Mvoxels/s' % (d['ms_per_step'], d['value']))" >> $out
python bench.py --config 5 --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config 5: %.2f ms/step' % d['ms_per_step'])" >> $out
done
cat $out
