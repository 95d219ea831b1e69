#!/bin/bash
# one timing line per workload (no CPU leg)
cd /root/repo
WL=${WL:-"c3 c4 c2 c2p c5 c1 pendulum:2001,2001:21:float32 cartpole:51,51,51,51:21:float32 pendulum:1001,1001:51:float64 cartpole:51,51,51,51:21:float64"}
for w in $WL; do
    st=20; wu=3
    case $w in c2|c2p|c1|pendulum*) st=2000; wu=200;; c4|c5) st=5; wu=2;; esac
    echo -n "$w: "
    timeout 300 python bench.py --workload $w --no-cpu --steps $st --warmup $wu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms setup %.0f ms'%(d['ms_per_step'], d['setup_ms']), d['kernel_path'][:150])"
done
