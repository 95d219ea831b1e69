#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fast3 or 3d_systems or remaining_dp_demos or self_check" > gpurun_out/r03_new_tests.log 2>&1
grep -E "^E  |passed|failed|Error" gpurun_out/r03_new_tests.log | head -30
python tools/tools_time.py h3 50 2>&1 | grep -E "TIME|nodes"
python tools/tools_time.py h3 50 NO_FAST=1 2>&1 | grep -E "TIME|nodes"
