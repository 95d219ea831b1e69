#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "second_form or f64 or edge_cases or slab or config1 or class_surface or mountaincar or mass or minimum_time or full_size_c2" 2>&1 | tail -4
for W in c1 pendulum:1001,1001:51:float64 pendulum:401,401:101:float64; do
  timeout 100 python tools/tools_ablate.py $W 100
done
