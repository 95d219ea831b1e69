#!/bin/bash
# round 4, first GPU call: the GPU suite on the renamed / compacted bench + kernel names, then the driver's default bench command
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r04_gputests_a.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_gputests_a.log
tail -12 gpurun_out/r04_gputests_a.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_a.stdout 2> gpurun_out/r04_bench_a.stderr ) 2>&1 | tail -4
wc -c gpurun_out/r04_bench_a.stdout
cat gpurun_out/r04_bench_a.stdout
grep -v "full record" gpurun_out/r04_bench_a.stderr | tail -20
