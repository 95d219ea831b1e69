#!/bin/bash
# A/B on one box: experiment builds of the lean unit (pyro_amd/libpyrovi_e<k>.so, built by hand with -DL4EXP=k) against the product
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_sched_flags.log; : > $L
for rep in 1 2; do
for lib in libpyrovi.so libpyrovi_e1.so libpyrovi_e2.so libpyrovi_e3.so libpyrovi_e4.so; do
  for w in "c3 200" "c4 40" "c2 2000" "c2p 2000"; do
    echo "== $lib $w" >> $L
    PYROVI_LIB=/root/repo/pyro_amd/$lib timeout 300 python tools/tools_time.py $w 2>&1 | grep -E "TIME|rror" | cut -c1-200 >> $L
  done
done
done
cat $L
