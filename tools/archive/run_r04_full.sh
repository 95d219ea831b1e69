#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r04_gputests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_gputests.log
tail -8 gpurun_out/r04_gputests.log
WL=${WL:-"c3 c4 c2 c2p c5 c5d c1 h3"} bash tools/tools_profile_r04.sh 2>&1 | tail -12
# the float32 accuracy mode (PVI_FLAG_F32_FEEDBACK, k_sweep_lean4fb): kernel-trace stats of its bench line
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04_stats_c3fb -o s -- python /root/repo/bench.py --workload c3 --f32-feedback --no-cpu --steps 10 --warmup 2 > /root/repo/gpurun_out/r04_bench_c3fb.json 2> /root/repo/gpurun_out/r04_bench_c3fb.err)
tail -c 300 gpurun_out/r04_bench_c3fb.json
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"
wc -c gpurun_out/r04_bench_final.json
