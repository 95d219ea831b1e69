#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or edge_cases or 4d_f32 or config1 or cartpole_21p4 or slab or full_size_c2 or randomised or two_ranks or world1 or class_surface" 2>&1 | tail -3
for W in c2 c2p c1; do
  PVI_SGPR100=1 timeout 60 python tools/tools_ablate.py $W 300
  timeout 60 python tools/tools_ablate.py $W 300
done
PVI_TV0=15 PVI_TV1=34 PVI_NO_SPLIT_FINISH=1 timeout 60 python tools/tools_ablate.py c3 10
PVI_TV0=15 PVI_TV1=34 timeout 60 python tools/tools_ablate.py c3 10
PVI_TV0=10 PVI_TV1=51 timeout 60 python tools/tools_ablate.py c3 10
timeout 60 python tools/tools_ablate.py c3 10
timeout 100 python tools/tools_ablate.py c4 5
