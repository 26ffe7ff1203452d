#!/bin/bash
# accuracy and time of the two-stage reduction over a few shapes: tools/band_probe_all.sh OUT
out=$1
for bw in "" 2 4; do
  for cfg in "1 512 64 volume" "4 512 64 volume" "32 512 64 volume" "32 512 64 graded" "8 256 32 random" "3 384 48 volume" "2 200 20 graded"; do
    if [ -z "$bw" ]; then unset NDMPS_TRD_BAND; else export NDMPS_TRD_BAND=$bw; fi
    timeout -k 10 120 python tools/band_probe.py $cfg >> $out 2>&1 || echo "band=$bw $cfg FAILED" >> $out
  done
done
cat $out
