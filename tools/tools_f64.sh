#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "second_form or f64 or tables or config1 or edge_cases or slab or class_surface or mountaincar or acrobot or generic or linear_state or minimum_time or full_size_c2 or randomised" 2>&1 | tail -4
for W in c1 pendulum:1001,1001:51:float64 c5 c5s cartpole:51,51,51,51:21:float64; do
  N=20; [ $W = c1 ] && N=300
  PVI_NO_SWEEP64=1 timeout 100 python tools/tools_ablate.py $W $N
  timeout 100 python tools/tools_ablate.py $W $N
done
