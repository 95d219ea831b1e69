#!/bin/bash
# parameter sweep of the tiled kernel: PVI_BLOCK x PVI_LDS_KB per workload
run() { python bench.py --workload $1 --no-cpu $2 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1 blk=$PVI_BLOCK lds=$PVI_LDS_KB split=$PVI_LSPLIT', 'kern_ms %.4f' % d['roofline']['kernel_ms'], 'Gcells/s %.1f' % (d['value']/1e9))
    elif 'rror' in l: print(l, end='')
"; }
for b in 128 256 512; do for k in 16 32 64; do PVI_BLOCK=$b PVI_LDS_KB=$k run c2; done; done
for s in 2 3 4 5; do PVI_LSPLIT=$s run c2p; done
for b in 256 512 1024; do for k in 64 96 150; do PVI_BLOCK=$b PVI_LDS_KB=$k run c3 "--steps 5 --warmup 1"; done; done
run c3 "--steps 5 --warmup 1"
