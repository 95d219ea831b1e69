#!/bin/bash
# round 4, error-feedback storage: drift curves on the GPU, randomised run
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_feedback.log; : > $L
timeout 600 python tools/tools_drift_fb.py cartpole:31,31,31,31:21:float32 600 50 >> $L 2>&1
timeout 1500 python tools/tools_drift_fb.py c3 2000 100 >> $L 2>&1
timeout 1100 python tools/tools_fuzz_fb.py 60 4242 > gpurun_out/r04_fuzz_fb.log 2>&1; tail -1 gpurun_out/r04_fuzz_fb.log >> $L
grep -E "DRIFT|feedback:" $L; grep FAIL gpurun_out/r04_fuzz_fb.log | cut -c1-220
