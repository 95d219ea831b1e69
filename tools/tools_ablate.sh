#!/bin/bash
# breakdown of the per-workgroup fixed cost of k_sweep_lean by switching parts off (timing only)
cd /root/repo
for W in c2 c3; do
  N=300; [ $W = c3 ] && N=10
  T=""; [ $W = c3 ] && T="PVI_TV0=15 PVI_TV1=34"
  for D in 0 4 12 2 16 18 30 32 62; do
    env $T PVI_DBG=$D timeout 60 python tools/tools_ablate.py $W $N
  done
done
