#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r03_stage.log; : > $L
python tools/tools_time.py c3 30 | grep -E "TIME|nodes" >> $L 2>&1
python tools/tools_time.py c3 30 TV0=19 TV1=26 | grep -E "TIME|nodes" >> $L 2>&1
cat $L
