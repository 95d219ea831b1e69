#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_align32.log; : > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree" > gpurun_out/r04_align32_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_align32_tests.log | tail -5 >> $L
for a in "ALIGN32=0" "ALIGN32=1" ""; do
  for w in "c3 200" "c4 40"; do
    echo "== $w $a" >> $L
    timeout 300 python tools/tools_time.py $w $a 2>&1 | grep -E "TIME|nodes|rror" | cut -c1-1500 >> $L
  done
done
cat $L
