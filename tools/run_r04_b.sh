#!/bin/bash
# A/B on one box: the library of the previous commit (pyro_amd/libpyrovi_prev.so, built by hand) against the current one
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_ab.log; : > $L
for rep in 1 2; do
for lib in pyro_amd/libpyrovi_prev.so pyro_amd/libpyrovi.so; do
  for w in "c3 200" "c4 40"; do
    echo "== $lib $w" >> $L
    PYROVI_LIB=/root/repo/$lib timeout 300 python tools/tools_time.py $w 2>&1 | grep -E "TIME|nodes|rror" | cut -c1-260 >> $L
  done
done
done
echo "== ORDER=1 c3" >> $L
timeout 300 python tools/tools_time.py c3 200 ORDER=1 2>&1 | grep -E "TIME|rror" >> $L
cat $L
