#!/bin/bash
cd /root/repo
for L in "" /root/repo/pyro_amd/libpyrovi_ds64.so; do
  echo "LIB=$L"
  PYROVI_LIB=$L timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or config1_solve_f32 or cartpole_21p4" 2>&1 | tail -2
  for W in c2 c2p; do PYROVI_LIB=$L timeout 60 python tools/tools_ablate.py $W 300 | cut -c1-60; done
  PYROVI_LIB=$L PVI_TV0=10 PVI_TV1=51 timeout 60 python tools/tools_ablate.py c3 10 | cut -c1-60
done
