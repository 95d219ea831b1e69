#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_sched_default.log; : > $L
timeout 900 python tools/tools_create_time.py c3 c4 repeats=3 2>&1 | grep -E "CREATE|rror" >> $L
timeout 300 python tools/tools_time.py c3s 200 2>&1 | grep -E "TIME|rror" >> $L
timeout 300 python tools/tools_time.py cartpole:61,61,61,61:21:float32 100 2>&1 | grep -E "TIME|rror" >> $L
cat $L
