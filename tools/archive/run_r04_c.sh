#!/bin/bash
# late round 4: randomised campaigns with fresh seeds on the final kernels
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_fuzz_late.log; : > $L
timeout 1500 python tools/tools_fuzz64.py 250 777 > gpurun_out/r04_fuzz64_late.log 2>&1; echo "fuzz64 250 seed 777: $(tail -1 gpurun_out/r04_fuzz64_late.log)" >> $L
timeout 1200 python tools/tools_fuzz.py 150 778 > gpurun_out/r04_fuzz32_late.log 2>&1; echo "fuzz32 150 seed 778: $(tail -1 gpurun_out/r04_fuzz32_late.log)" >> $L
timeout 900 python tools/tools_fuzz_tiers.py 80 779 > gpurun_out/r04_fuzztiers_late.log 2>&1; echo "fuzz tiers 80 seed 779: $(tail -1 gpurun_out/r04_fuzztiers_late.log)" >> $L
cat $L
