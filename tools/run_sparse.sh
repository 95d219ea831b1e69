#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "second_form or self_check" 2>&1 | tail -3
for w in c5 c5s twolink:51,51,51,51:11,11:float64 twolink:71,71,71,71:11,11:float64; do
    echo -n "$w: "
    timeout 300 python bench.py --workload $w --no-cpu --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms setup %.0f ms'%(d['ms_per_step'], d['setup_ms']), d['kernel_path'])"
done
