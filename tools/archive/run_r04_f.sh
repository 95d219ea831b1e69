#!/bin/bash
# the feedback kernel's bench line under rocprofv3 (kernel-trace stats), and the plain kernel's on the same box
cd /root/repo; mkdir -p gpurun_out; rm -rf gpurun_out/r04_stats_c3fb gpurun_out/r04_stats_c3same
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04_stats_c3fb -o s -- python /root/repo/bench.py --workload c3 --f32-feedback --no-cpu --steps 10 --warmup 2 > /root/repo/gpurun_out/r04_bench_c3fb.json 2> /root/repo/gpurun_out/r04_bench_c3fb.err)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04_stats_c3same -o s -- python /root/repo/bench.py --workload c3 --no-cpu --steps 10 --warmup 2 > /root/repo/gpurun_out/r04_bench_c3same.json 2> /root/repo/gpurun_out/r04_bench_c3same.err)
for d in c3fb c3same; do f=$(find gpurun_out/r04_stats_$d -name "*kernel_stats.csv" | head -1); head -2 $f | tail -1 | cut -c1-60; head -2 $f | tail -1 | awk -F, '{print $(NF-6), $(NF-4)}'; done
