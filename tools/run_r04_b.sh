#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_persist.log; : > $L
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree or 4d_f32 or cartpole_21p4" 2>&1 | tail -5 >> $L
for w in c3 c4; do
  n=30; [ $w = c4 ] && n=10
  python tools/tools_time.py $w $n 2>&1 | grep -E "TIME|nodes|rror" >> $L
done
cat $L
