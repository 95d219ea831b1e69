#!/bin/bash
# round 5: first GPU call after the boxes reopen -- everything written blind, safest first, each under its own timeout
cd /root/repo; mkdir -p gpurun_out; O=gpurun_out/r05_validate; mkdir -p $O
t() { # name timeout pytest-args...
  local n=$1 to=$2; shift 2
  timeout $to python -m pytest "$@" -m gpu -q -x > $O/$n.log 2>&1; echo "$n rc=$? :: $(tail -1 $O/$n.log)" | tee -a $O/summary.log
}
: > $O/summary.log
t stress 600 tests/test_gpu_stress.py -k "not 2d_float32"
t variants 900 tests/test_gpu_parity.py -k "variants or self_check"
t fb4 900 tests/test_gpu_parity.py -k "feedback_storage_small or two_ranks_share_one_gpu"
t fb2 900 tests/test_gpu_parity.py -k "feedback_storage_on_2d or node_table_tier or explicit_system"
t swapped 900 tests/test_gpu_parity.py -k "swapped_internal_order"
bash tools/run_r05_c.sh > $O/ab.log 2>&1; tail -20 $O/ab.log
cat $O/summary.log
