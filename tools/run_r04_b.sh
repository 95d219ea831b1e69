#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_quad_resched.log; : > $L
for a in "WIN=2" "WIN=3" "WIN=2 TV0=20 TV1=20" "PERSIST=1"; do
  timeout 300 python tools/tools_time.py c3 30 $a 2>&1 | grep -E "TIME|rror" >> $L
done
timeout 300 python tools/tools_time.py c4 10 WIN=2 2>&1 | grep -E "TIME|rror" >> $L
cat $L
