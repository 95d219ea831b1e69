#!/bin/bash
run() { python bench.py --workload $1 --no-cpu --steps ${2:-100} --warmup 5 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1 $3', 'kern_us %.1f' % (d['roofline']['kernel_ms']*1e3), 'Gcells/s %.1f' % (d['value']/1e9), d['kernel_path'][:90])
    elif 'rror' in l: print(l, end='')
"; }
PVI_GUARD=0 run c2 100 guard0
PVI_GUARD=1e-4 run c2 100 guard1e-4
for t in 64 128 192 251 334 501; do PVI_TV0=1 PVI_TV1=$t PVI_LDS_KB=60 run c2 100 tv1=$t; done
PVI_GUARD=0 run c3 5 guard0
for sh in "8 26" "10 26" "15 26" "19 26" "6 51" "10 51" "4 101" ; do set -- $sh; PVI_TV0=$1 PVI_TV1=$2 PVI_LDS_KB=150 run c3 5 "tile=$1x$2"; done
