#!/bin/bash
cd /root/repo
for sh in "15 34" "12 34" "10 34" "8 34" "7 34" "10 51" "8 51" "6 51" "5 51" "20 26" "14 26" "10 26"; do
  set -- $sh
  echo -n "tile $1x$2: "
  PVI_TV0=$1 PVI_TV1=$2 timeout 120 python bench.py --workload c3 --no-cpu --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms'%(d['ms_per_step']), d['kernel_path'][:110])"
done
