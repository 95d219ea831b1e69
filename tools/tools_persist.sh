#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or config1_solve_f32 or edge_cases or full_size_c2 or mountaincar or slab or two_ranks or self_check or mass or minimum_time or class_surface" 2>&1 | tail -3
for W in c2 c2p pendulum:401,401:101:float32 pendulum:2001,2001:21:float32; do timeout 60 python tools/tools_ablate.py $W 300 | cut -c1-110; done
