#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r04_gputests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_gputests.log
tail -8 gpurun_out/r04_gputests.log
WL=${WL:-"c3 c4 c2 c2p c5 c5d c1 h3"} bash tools/tools_profile_r04.sh 2>&1 | tail -12
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"
wc -c gpurun_out/r04_bench_final.json
