#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "policy_evaluator or explicit_systems or rollout or variants_agree or sanitized" > gpurun_out/r03_new_tests.log 2>&1
grep -E "^E  |passed|failed|Error" gpurun_out/r03_new_tests.log | head -30
