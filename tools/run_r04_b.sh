#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_c5_chunks2.log; : > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "second_form or sweeps_f64 or fast3 or multi_sweep or full_size_sampled" 2>&1 | tail -4 >> $L
for w in c5 c5d c5s c5ds "cartpole:51,51,51,51:21:float64"; do
for a in "" "XCD_CHUNK=0"; do
  n=10; [ $w = c5s ] && n=100; [ $w = c5ds ] && n=100
  timeout 300 python tools/tools_time.py $w $n $a 2>&1 | grep -E "TIME|rror" >> $L
done; done
cat $L
