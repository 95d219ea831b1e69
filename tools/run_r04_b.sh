#!/bin/bash
# round 4: persistent 4-D sweep against one tile per workgroup
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_persist.log; : > $L
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree" 2>&1 | tail -5 >> $L
for w in c3 c4; do
  n=30; [ $w = c4 ] && n=10
  python tools/tools_time.py $w $n PERSIST=0 2>&1 | grep -E "TIME|nodes|rror" >> $L
  python tools/tools_time.py $w $n PERSIST=1 2>&1 | grep -E "TIME|nodes|rror" >> $L
  python tools/tools_time.py $w $n PERSIST=1 PERSIST_WGS=2 2>&1 | grep -E "TIME|rror" >> $L
done
cat $L
