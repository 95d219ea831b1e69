#!/bin/bash
# one compact bench line: tools_b1.sh <workload> [steps] [warmup] (env passes through)
python bench.py --workload $1 --no-cpu --steps ${2:-20} --warmup ${3:-2} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', 'ms/sweep %.4f' % d['ms_per_step'], 'kern_ms %.4f' % d['roofline']['kernel_ms'], 'Gcells/s %.1f' % (d['value']/1e9), d.get('kernel_path','')[:100])
    elif 'rror' in l or 'Trace' in l: print(l, end='')
"
