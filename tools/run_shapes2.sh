#!/bin/bash
cd /root/repo
for sh in "8 32" "16 32" "4 64" "8 64" "2 128" "4 128" "32 32" "16 64"; do
  set -- $sh
  echo -n "tile $1x$2: "
  PVI_NPT=1 PVI_TV0=$1 PVI_TV1=$2 timeout 120 python bench.py --workload c2 --no-cpu --steps 2000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us'%(d['ms_per_step']*1e3), d['kernel_path'][:100])"
done
