#!/bin/bash
for c in metric 4; do
python tools/scratch/gc_probe.py --config $c --steps 200 --skip-single --no-cpu-baseline --no-configs 2>&1 >/dev/null | grep "^#" | python -c "
import sys
for l in sys.stdin:
    if l.startswith('# host time per step'):
        t=[float(x) for x in l.split(':')[1].split()]
        t=t[2:]
        print('config $c: steps', len(t), 'max host ms %.1f' % max(t), 'mean %.2f' % (sum(t)/len(t)), 'steps > 100 ms:', sum(x>100 for x in t))
    elif l.startswith('# 200 steps'): print(l[:40])
"
done
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
