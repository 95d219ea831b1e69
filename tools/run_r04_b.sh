#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_deferred64.log; : > $L
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config1 or c2_solve or second_form or sweeps_f64 or class_surface or multi_sweep or edge_cases or c3_full_size_solved or full_size_c2 or mountaincar or acrobot or floatmass or mintime" > gpurun_out/r04_deferred64_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_deferred64_tests.log | tail -6 >> $L
for w in pendulum:201,201:21:float64 pendulum:401,401:51:float64 pendulum:1001,1001:51:float64 c5s; do for a in "" "DEFER=0"; do
  n=2000; [ $w = c5s ] && n=200
  timeout 300 python tools/tools_time.py $w $n $a 2>&1 | grep -E "TIME|rror" >> $L
done; done
cat $L
