#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_rtm.log; : > $L
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_sweep or config1 or golden or pendulum or c1" > gpurun_out/r04_rtm_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_rtm_tests.log | tail -8 >> $L
timeout 900 python tools/tools_fuzz64.py 100 41 > gpurun_out/r04_fuzz64d.log 2>&1; tail -1 gpurun_out/r04_fuzz64d.log >> $L
for w in "c1" "pendulum:201,201:21:float64" "pendulum:201,201:11:float64" "pendulum:101,101:21:float64" "pendulum:51,401:24:float64" "pendulum:101,101:41:float64" "cartpole:9,9,9,9:5:float64"; do
for a in "" "REGTAB=0" "MULTI=0"; do
  echo "== $w $a" >> $L
  timeout 300 python tools/tools_time.py $w 3000 $a 2>&1 | grep -E "TIME|rror" | cut -c1-260 >> $L
done
done
cat $L
