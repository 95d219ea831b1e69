#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r03_align.log; : > $L
python tools/tools_time.py c3 30 | grep -E "TIME|nodes" >> $L 2>&1
python tools/tools_time.py c3 30 ALIGN=0 | grep -E "TIME|nodes" >> $L 2>&1
python tools/tools_time.py c4 10 | grep -E "TIME|nodes" >> $L 2>&1
cat $L
export PVI_ROUND=r03; bash tools/tools_counters.sh c3 > gpurun_out/r03_counters_c3.log 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree or sampled_against_c_oracle" > gpurun_out/r03_l4_variants.log 2>&1; grep -E "^E  |passed|failed" gpurun_out/r03_l4_variants.log | head -20
