run() { python bench.py --workload $1 --no-cpu --steps ${2:-50} --warmup 3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', 'kern_us %.1f' % (d['roofline']['kernel_ms']*1e3), 'Gcells/s %.1f' % (d['value']/1e9))
    elif 'rror' in l: print(l, end='')
"; }
for A in 1 3 11 27 51 99 203; do PVI_LSPLIT=0 run pendulum:1001,1001:$A:float32; done
for A in 1 5 21 41 81; do PVI_LSPLIT=0 run cartpole:101,101,101,101:$A:float32 6; done
