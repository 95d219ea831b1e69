#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_quad2.log; : > $L
for args in "TV0=20 TV1=20" "TV0=20 TV1=20 RSMOD=0" "TV0=20 TV1=20 RSMOD=1" "TV0=20 TV1=20 RSMOD=8" "TV0=20 TV1=20 NO_XCD=1" "TV0=10 TV1=20" "TV0=10 TV1=20 RSMOD=0" "TV0=10 TV1=10" "TV0=10 TV1=10 RSMOD=0" "TV0=5 TV1=20"; do
  timeout 300 python tools/tools_time.py c3 20 WIN=2 $args 2>&1 | grep -E "TIME|rror" >> $L
done
cat $L
