#!/bin/bash
cd /root/repo
for lib in libpyrovi_w3.so libpyrovi.so libpyrovi_w5.so; do
  for w in c5 c5s twolink:71,71,71,71:11,11:float64; do
    echo -n "$lib $w: "
    PYROVI_LIB=/root/repo/pyro_amd/$lib timeout 300 python bench.py --workload $w --no-cpu --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms setup %.0f ms'%(d['ms_per_step'], d['setup_ms']), d['kernel_path'])"
  done
done
