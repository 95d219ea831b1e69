#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r03_stage.log; : > $L
python tools/tools_time.py c3 30 | grep -E "TIME|nodes" >> $L 2>&1
cat $L
WL="c3 c2" bash tools/tools_profile_r03.sh 2>&1 | tail -5
head -5 gpurun_out/r03_stats_c3/by_launch.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree or sampled_against_c_oracle" > gpurun_out/r03_l4_variants.log 2>&1; grep -E "^E  |passed|failed" gpurun_out/r03_l4_variants.log | head -20
