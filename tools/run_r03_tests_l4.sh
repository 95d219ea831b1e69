#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree" > gpurun_out/r03_l4_variants.log 2>&1
tail -40 gpurun_out/r03_l4_variants.log
