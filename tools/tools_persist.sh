#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or edge_cases or 4d_f32 or config1 or cartpole_21p4 or slab or full_size or randomised or two_ranks" 2>&1 | tail -3
for W in c2 c3 c2p; do
  N=300; [ $W = c3 ] && N=10
  T=""; [ $W = c3 ] && T="PVI_TV0=15 PVI_TV1=34"
  env $T PVI_DBG=128 timeout 60 python tools/tools_ablate.py $W $N
  env $T timeout 60 python tools/tools_ablate.py $W $N
  env $T PVI_PERSIST=1 timeout 60 python tools/tools_ablate.py $W $N
done
timeout 100 python tools/tools_ablate.py c4 5
