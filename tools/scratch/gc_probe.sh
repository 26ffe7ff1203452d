#!/bin/bash
out=gpurun_out/gc_probe.txt
: > $out
for cfg in 5 metric; do
  for fr in "" 1 "" 1; do
    echo "== config $cfg GCFREEZE=$fr" >> $out
    GCFREEZE=$fr python tools/scratch/gc_probe.py --config $cfg --skip-single --no-cpu-baseline --no-configs 2>&1 >/dev/null | grep "^#" >> $out
  done
done
cat $out
