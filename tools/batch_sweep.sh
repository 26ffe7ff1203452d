#!/bin/bash
# Throughput of bench.py over batch / group sizes (one line each): tools/batch_sweep.sh OUT "64 2" "56 2" ...
out=$1; shift
for cfg in "$@"; do
  set -- $cfg
  python bench.py --batch $1 --groups $2 --skip-single --no-cpu-baseline --steps 10 --warmup 3 > /tmp/bs.json 2> /tmp/bs.err || { echo "batch $1 groups $2 FAILED: $(tail -2 /tmp/bs.err)" >> $out; continue; }
  python -c "import sys,json; d=json.load(open('/tmp/bs.json')); print('batch', d['config']['batch_per_gpu'], 'groups', d['config']['groups_per_gpu'], 'Mvox/s %.0f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'ms/vol %.3f' % (d['ms_per_step']/d['config']['batch_per_gpu']))" >> $out
done
cat $out
