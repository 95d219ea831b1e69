#!/bin/bash
# A/B on one box: the library before (e0) and after (e1) the exact-pass lanes moved to a scalar mask and J_k(self) is read behind the loops
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_tail_spills.log; : > $L
for rep in 1 2 3; do
for lib in libpyrovi_e0.so libpyrovi_e1.so; do
  for w in c3 c4; do
  n=200; [ $w = c4 ] && n=40
  echo "== $lib $w" >> $L
  PYROVI_LIB=/root/repo/pyro_amd/$lib timeout 300 python tools/tools_time.py $w $n 2>&1 | grep -E "TIME|rror" | cut -c1-200 >> $L
  done
done
done
cat $L
