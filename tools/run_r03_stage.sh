#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r03_rs32.log; : > $L
python tools/tools_time.py c3 30 | grep -E "TIME|nodes" >> $L 2>&1
python tools/tools_time.py c4 10 | grep -E "TIME|nodes" >> $L 2>&1
python tools/tools_time.py c4 10 NO_RS64=1 | grep -E "TIME|nodes" >> $L 2>&1
cat $L
