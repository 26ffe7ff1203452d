#!/bin/bash
out=gpurun_out/invit_ablate.txt
: > $out
for cb in 128 64 32 16; do
for dbg in 0 7; do
  echo "== NDMPS_INVIT_CB=$cb NDMPS_INVIT_DBG=$dbg (1: no chain, 2: no helper loads, 4: no write-back)" >> $out
  NDMPS_INVIT_CB=$cb NDMPS_INVIT_DBG=$dbg python tools/trd_probe.py 32 512 64 2>&1 | grep "invit phases\|B=" | cut -c1-140 >> $out
  NDMPS_INVIT_CB=$cb NDMPS_INVIT_DBG=$dbg python tools/trd_probe.py 1 512 64 2>&1 | grep "invit phases\|B=" | cut -c1-140 >> $out
done
done
cat $out
