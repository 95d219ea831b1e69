#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_fill128.log; : > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree or cartpole or twolink" > gpurun_out/r04_fill128_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_fill128_tests.log | tail -5 >> $L
for w in "c3 200" "c4 40"; do
  echo "== $w" >> $L
  timeout 300 python tools/tools_time.py $w 2>&1 | grep -E "TIME|rror" | cut -c1-300 >> $L
done
bash tools/tools_ldsconf.sh c3 c3_f128 TV0=19 TV1=51 >> $L 2>&1
bash tools/tools_ldsconf.sh c4 c4_f128 TV0=55 TV1=26 >> $L 2>&1
cat $L
