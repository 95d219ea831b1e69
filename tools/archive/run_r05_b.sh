#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash tools/run_r05_ab.sh step3 libpyrovi_base.so libpyrovi.so c3 c4 > /dev/null 2>&1
grep -E "TIME|==" gpurun_out/r05_ab_step3.log | paste - - | awk '{print $2,$3,$(NF-1)}'
timeout 1500 python -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py -m gpu -q -x -k "stress or determin or lockstep or variants or cartpole or full_size_sampled or feedback_storage_small or self_check or shard_schedule" > gpurun_out/r05_b_tests.log 2>&1; tail -5 gpurun_out/r05_b_tests.log
