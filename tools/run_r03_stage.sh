#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r03_c2p_lsplit.log; : > $L
for ls in "" LSPLIT=1 LSPLIT=2 LSPLIT=3 LSPLIT=4 LSPLIT=5; do
python tools/tools_time.py c2p 2000 $ls | grep -E "TIME|nodes" | cut -c1-260 >> $L 2>&1
done
for ls in "" LSPLIT=1 LSPLIT=2 LSPLIT=3; do
python tools/tools_time.py pendulum:101,101:11:float32 2000 $ls | grep -E "TIME|nodes" | cut -c1-260 >> $L 2>&1
done
cat $L
