#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "second_form or f64 or edge_cases or slab or (full_size_sampled and c5) or caller_transport or world1" 2>&1 | tail -4
for W in c5 cartpole:51,51,51,51:21:float64 twolink:41,41,41,41:5,5:float64; do
  timeout 100 python tools/tools_ablate.py $W 10
done
