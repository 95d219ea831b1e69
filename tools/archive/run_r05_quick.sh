#!/bin/bash
# a GPU window of a quarter of an hour: the determinism / lockstep stress tests, the swapped-order test, one timing each of C3 as
# built, swapped, and with the uniform-displacement ablation
cd /root/repo; mkdir -p gpurun_out; O=gpurun_out/r05_quick.log; : > $O
timeout 400 python -m pytest tests/test_gpu_stress.py -m gpu -q -x -k "not 2d_float32" 2>&1 | tail -2 >> $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "swapped_internal_order" 2>&1 | tail -3 >> $O
for ov in "" "ORDER=swapped" "ORDER=swapped TV0=15 TV1=34 RS_CONG=1"; do timeout 120 python tools/tools_time.py c3 200 $ov 2>&1 | grep TIME | cut -c1-140 >> $O; done
PYROVI_LIB=/root/repo/pyro_amd/libpyrovi_ablate.so timeout 120 python tools/tools_time.py c3 200 2>&1 | grep TIME | sed 's/^/ablate-uniform /' | cut -c1-140 >> $O
cat $O
