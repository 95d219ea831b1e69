#!/bin/bash
# round 3, first GPU pass: the new parity tests + the default bench line in the new format
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=10 -k "not variants_agree" > gpurun_out/r03_gputests_first.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_gputests_first.log
tail -25 gpurun_out/r03_gputests_first.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "solved_to_tolerance" > gpurun_out/r03_c3_converged.log 2>&1
grep -E "sweeps|after|passed|failed" gpurun_out/r03_c3_converged.log | tail -25
python bench.py > gpurun_out/r03_bench_first.json 2> gpurun_out/r03_bench_first.err
echo "bench rc=$?"; tail -c 1500 gpurun_out/r03_bench_first.err; head -c 1200 gpurun_out/r03_bench_first.json
