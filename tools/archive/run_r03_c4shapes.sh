#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r03_c4_shapes.log; : > $L
for s in "21 22" "21 24" "23 22" "25 20" "16 32" "12 42" "30 17" "38 13" "21 18" "17 22"; do set -- $s
python tools/tools_time.py c4 8 TV0=$1 TV1=$2 | grep -E "TIME|nodes" | sed -e 's/lsplit.*sparse=0//' >> $L 2>&1
done
cat $L
