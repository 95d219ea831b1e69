#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_quad3.log; : > $L
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree" 2>&1 | tail -8 >> $L
for w in c3 c4; do
  n=30; [ $w = c4 ] && n=10
  timeout 300 python tools/tools_time.py $w $n WIN=3 2>&1 | grep -E "TIME|nodes|rror" >> $L
done
cat $L
