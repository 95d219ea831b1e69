#!/bin/bash
# whole GPU test suite (no -x: every test reports), then the N > 1 bench harness with one rank
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r02_gputests_suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_gputests_suite.log
tail -8 gpurun_out/r02_gputests_suite.log
PVI_FORCE_PARALLEL=1 timeout 600 python bench.py --gpus 1 > gpurun_out/r02_bench_world1.json 2> gpurun_out/r02_bench_world1.err
echo "bench world1 rc=$?"; tail -c 1500 gpurun_out/r02_bench_world1.json; tail -5 gpurun_out/r02_bench_world1.err
