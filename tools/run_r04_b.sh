#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_c4_bands_traffic.log; : > $L
for b in 2 3 4 6; do
  timeout 300 python tools/tools_time.py c4 10 TV0=64 TV1=27 BANDS=$b 2>&1 | grep -E "TIME|rror" >> $L
  bash tools/tools_traffic_quick.sh c4 TV0=64 TV1=27 BANDS=$b 2>&1 | grep -E "TRAFFIC|rror" >> $L
done
cat $L
