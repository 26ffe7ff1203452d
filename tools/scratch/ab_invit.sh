#!/bin/bash
# A/B of the inverse iteration's column-block width (NDMPS_INVIT_CB) and of the side stream of the T factors
# (NDMPS_NO_SIDE_STREAM) on one or a few big matrices, every eigenpair at order 4096 and BASELINE config 5.
out=gpurun_out/ab_invit.txt
: > $out
for cb in 128 default; do
  if [ $cb = default ]; then unset NDMPS_INVIT_CB; else export NDMPS_INVIT_CB=$cb; fi
  echo "== NDMPS_INVIT_CB=$cb" >> $out
  python tools/panel_probe.py 1024,2048 128 1 2>&1 | grep panel >> $out
  python tools/panel_probe.py 1024 128 4 2>&1 | grep panel >> $out
  python tools/full_probe.py 2048,4096 --no-jacobi >> $out 2>&1
done
unset NDMPS_INVIT_CB
for v in "NDMPS_INVIT_CB=128" "NDMPS_NO_SIDE_STREAM=1" "A=1" "NDMPS_INVIT_CB=32" "NDMPS_INVIT_CB=128" "A=1"; do
  env $v python bench.py --config 5 --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config 5 $v: %.2f ms/step' % d['ms_per_step'])" >> $out
done
cat $out
