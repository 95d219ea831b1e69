#!/bin/bash
# quick GPU check: parity tests + short bench lines (compact)
python -m pytest tests -m gpu -q -x 2>&1 | tail -5
for w in c2 c2p c3; do
  python bench.py --workload $w --no-cpu $( [ $w = c3 ] && echo "--steps 10 --warmup 2" ) 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['config']['workload'][:4], 'ms/sweep %.4f' % d['ms_per_step'], 'kern_ms %.4f' % d['roofline']['kernel_ms'], 'Gcells/s %.1f' % (d['value']/1e9), 'sweeps/s %.1f' % d['sweeps_per_sec'], d.get('kernel_path',''))
    else: print(l, end='')
"
done
