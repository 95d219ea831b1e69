#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r03_stage.log; : > $L
python tools/tools_time.py c3 30 | grep -E "TIME|nodes" >> $L 2>&1
python tools/tools_time.py c4 10 | grep -E "TIME|nodes" >> $L 2>&1
cat $L
export PVI_ROUND=r03; for w in c3 c4; do bash tools/tools_counters.sh $w > gpurun_out/r03_counters_$w.log 2>&1; done
