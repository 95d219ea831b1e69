#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nearest or sharded_policy or policy_evaluator or spline or interpol or table_tier" 2>&1 | tail -15
