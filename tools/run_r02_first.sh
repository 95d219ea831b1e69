#!/bin/bash
# first GPU pass of round 2: tests, default bench, CPU scaling of the oracle twin
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02_gputests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_gputests.log
tail -30 gpurun_out/r02_gputests.log
python tools/cpu_scaling.py > gpurun_out/r02_cpu_scaling.json 2> gpurun_out/r02_cpu_scaling.err
cat gpurun_out/r02_cpu_scaling.json
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r02_bench_default.json
tail -5 gpurun_out/r02_bench_default.err
