#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_regtab2.log; : > $L
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_sweep or config1 or golden or pendulum or c1" > gpurun_out/r04_regtab2_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_regtab2_tests.log | tail -8 >> $L
for a in "MULTI=0" "REGTAB=0" "JWIN=0" ""; do
  echo "== c1 $a" >> $L
  timeout 300 python tools/tools_time.py c1 2000 $a 2>&1 | grep -E "TIME|rror" | cut -c1-300 >> $L
done
cat $L
