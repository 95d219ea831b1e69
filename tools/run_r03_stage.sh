#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r03_stage.log; : > $L
python tools/tools_time.py c3 30 | grep -E "TIME|nodes" >> $L 2>&1
python tools/tools_time.py c3 30 STAGE=1 | grep -E "TIME|nodes" >> $L 2>&1
for s in "10 17" "10 21" "7 34" "11 34" "15 34"; do set -- $s
  for st in 2 1; do
    python tools/tools_time.py c3 30 TV0=$1 TV1=$2 STAGE=$st | grep TIME >> $L 2>&1
  done
done
python tools/tools_time.py c4 10 | grep -E "TIME|nodes" >> $L 2>&1
cat $L
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree" > gpurun_out/r03_l4_variants.log 2>&1; grep -E "^E  |passed|failed" gpurun_out/r03_l4_variants.log | head -20
