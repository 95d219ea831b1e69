#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r03_ablate.log; : > $L
python tools/tools_time.py c3 30 | grep -E "TIME|nodes" >> $L 2>&1
for s in "7 34" "11 34" "10 51"; do set -- $s
  for d in 0 1 3 7; do
    python tools/tools_time.py c3 30 TV0=$1 TV1=$2 L4DBG=$d | grep TIME >> $L 2>&1
  done
done
python tools/tools_time.py c4 10 | grep -E "TIME|nodes" >> $L 2>&1
cat $L
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree" 2>&1 | tail -3
