#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_c2p_sact.log; : > $L
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config1 or variants_agree or north_star or edge_cases or class_surface or lowdef or demo or c2_solve" > gpurun_out/r04_sact_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_sact_tests.log | tail -5 >> $L
for a in "" "LSPLIT=2" "LSPLIT=4" "LSPLIT=3 TV0=2 TV1=16"; do
  timeout 300 python tools/tools_time.py c2p 4000 $a 2>&1 | grep -E "TIME|rror" >> $L
done
timeout 300 python tools/tools_time.py pendulum:101,101:51:float32 4000 2>&1 | grep -E "TIME|nodes|rror" | cut -c1-200 >> $L
timeout 300 python tools/tools_time.py pendulum:401,401:21:float32 4000 2>&1 | grep -E "TIME|rror" >> $L
cat $L
