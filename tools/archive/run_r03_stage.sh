#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r03_stage.log; : > $L
python tools/tools_time.py c3 30 | grep -E "TIME|nodes" >> $L 2>&1
python tools/tools_time.py c4 10 | grep -E "TIME|nodes" >> $L 2>&1
python tools/tools_time.py cartpole:21,21,21,21:7:float32 10 | grep -E "TIME|nodes" >> $L 2>&1
cat $L
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree or sampled_against_c_oracle or world1 or shard" > gpurun_out/r03_l4_variants.log 2>&1; grep -E "^E  |passed|failed" gpurun_out/r03_l4_variants.log | head -20
