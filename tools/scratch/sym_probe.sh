#!/bin/bash
out=gpurun_out/sym_probe.txt
: > $out
for b in 32 64; do
for sym in "" 1; do
echo "== B=$b NDMPS_TRD_SYM=$sym" >> $out
NDMPS_TRD_SYM=$sym python tools/trd_probe.py $b 512 64 2>&1 | grep "B=" | cut -c1-120 >> $out
done
done
for sym in 0 1; do
NDMPS_TRD_SYM=$sym python bench.py --skip-single --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline NDMPS_TRD_SYM=$sym: %.2f ms/step %.0f Mvoxels/s' % (d['ms_per_step'], d['value']))" >> $out
done
cat $out
