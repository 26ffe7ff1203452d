#!/bin/bash
# A/B of an environment switch on the timed loop of bench.py: tools/ab_bench.sh VAR [runs]
# prints Mvoxels/s, ms per step and the in-situ roofline fraction of the Gram kernel with VAR unset and VAR=1
var=$1; runs=${2:-2}
for r in $(seq $runs); do
  for mode in off on; do
    if [ $mode = on ]; then export $var=1; else unset $var; fi
    python bench.py --skip-single --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var $mode: %.0f Mvoxels/s  %.2f ms/step  gram frac %.3f (%.0f us)  team %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_us'], d['roofline']['tridiagonalisation']['launch_ms']))"
  done
done
