#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_multi.log; : > $L
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_sweep or config1 or class_surface_config1" 2>&1 | tail -15 >> $L
for ov in "" "MULTI=0"; do
  timeout 300 python tools/tools_time.py c1 4000 $ov 2>&1 | grep -E "TIME|nodes|rror" >> $L
  timeout 300 python tools/tools_time.py pendulum:51,51:9:float64 4000 $ov 2>&1 | grep -E "TIME|rror" >> $L
  timeout 300 python tools/tools_time.py pendulum:121,121:21:float64 2000 $ov 2>&1 | grep -E "TIME|rror" >> $L
done
cat $L
