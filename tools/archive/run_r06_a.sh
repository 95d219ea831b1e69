#!/bin/bash
# round 6, first GPU call: does HEAD's library (every cart-pole code object re-rolled since the last GPU run, 26 kernels never run)
# compute the right thing?  Safest first, each stage under its own timeout, no -x inside a stage (every failure is wanted).
cd /root/repo; mkdir -p gpurun_out/r06_a; O=gpurun_out/r06_a; : > $O/summary.log
t() { local n=$1 to=$2; shift 2
  timeout $to python -m pytest "$@" -m gpu -q > $O/$n.log 2>&1; echo "$n rc=$? :: $(tail -1 $O/$n.log)" | tee -a $O/summary.log; }
t stress 600 tests/test_gpu_stress.py
t parity 1800 tests/test_gpu_parity.py --durations=8
t zz_swapped 600 tests/test_gpu_zz_unproven.py -k swapped
t zz_fb2d 900 tests/test_gpu_zz_unproven.py -k "feedback_storage_on_2d"
t zz_fbexp 600 tests/test_gpu_zz_unproven.py -k "explicit_system or node_table_tier"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.log
t zz_multi32 300 tests/test_gpu_zz_unproven.py -k "2d_float32"
cat $O/summary.log
