#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_fuzz_final.log; : > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree or h3_full" > gpurun_out/r04_pin_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_pin_tests.log | tail -5 >> $L
timeout 1500 python tools/tools_fuzz64.py 160 11 > gpurun_out/r04_fuzz64.log 2>&1; tail -2 gpurun_out/r04_fuzz64.log >> $L
grep -c "regtab=1" gpurun_out/r04_fuzz64.log >> $L
FUZZ_4D_MAX=18 timeout 1500 python tools/tools_fuzz.py 80 12 > gpurun_out/r04_fuzz32.log 2>&1; tail -2 gpurun_out/r04_fuzz32.log >> $L
cat $L
