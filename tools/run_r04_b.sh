#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_final_check.log; : > $L
timeout 300 python tools/tools_time.py c4 10 2>&1 | grep -E "TIME|nodes|rror" | cut -c1-400 >> $L
timeout 300 python tools/tools_time.py c3 40 2>&1 | grep -E "TIME|rror" >> $L
timeout 300 python tools/tools_time.py c4 10 2>&1 | grep -E "TIME|rror" >> $L
timeout 300 python tools/tools_time.py c3 40 2>&1 | grep -E "TIME|rror" >> $L
bash tools/tools_traffic_quick.sh c4 2>&1 | grep -E "TRAFFIC|rror" >> $L
cat $L
