#!/bin/bash
cd /root/repo
for w in c2 pendulum:2001,2001:21:float32 pendulum:1001,1001:21:float32 pendulum:1001,1001:35:float32 pendulum:801,801:101:float32; do
  for t in 1 0; do
    echo -n "tune=$t $w: "
    PVI_TUNE=$t timeout 120 python bench.py --workload $w --no-cpu --steps 2000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us setup %.0f ms'%(d['ms_per_step']*1e3, d['setup_ms']), d['kernel_path'][10:24], d['kernel_path'][110:130])"
  done
done
