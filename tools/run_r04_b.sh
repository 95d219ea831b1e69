#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config1 or c2_solve or variants_agree or full_size_c2 or north_star or edge_cases or class_surface or lowdef or demo or two_ranks or sharded_driver" > gpurun_out/r04_deferred_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_deferred_tests.log | tail -8
