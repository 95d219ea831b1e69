#!/bin/bash
# round-4 profiles: for every workload (a) four rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | two SQ sets, kernel-trace
# only) -> gpurun_out/counters_<w>.json with the kernel variant they were taken with, (b) kernel-trace stats of
# `bench.py --workload <w> --no-cpu`; plus kernel-trace stats of the default `python bench.py`.
cd /root/repo
export PVI_ROUND=r04
WL=${WL:-"c3 c4 c2 c2p c5 c5d c1 h3"}
for w in $WL; do
  bash tools/tools_counters.sh $w > gpurun_out/r04_counters_$w.log 2>&1
  S=""; ([ $w = c3 ] || [ $w = c4 ] || [ $w = c5 ] || [ $w = c5d ]) && S="--steps 10 --warmup 2"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04_stats_$w -o s -- python /root/repo/bench.py --workload $w --no-cpu $S > /root/repo/gpurun_out/r04_stats_$w.log 2>&1)
  tail -1 gpurun_out/r04_stats_$w.log | cut -c1-160
done
python tools/make_counters_json.py $WL
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04_stats_default -o s -- python /root/repo/bench.py > /root/repo/gpurun_out/r04_bench_default.json 2> /root/repo/gpurun_out/r04_bench_default.err)
tail -c 400 gpurun_out/r04_bench_default.json
# the GPU box merges only gpurun_out/ back: tools/collect_profiles.sh (run in the repository afterwards) copies the summaries
# into profiles/
for w in $WL default; do
  t=$(find gpurun_out/r04_stats_$w -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python tools/kernel_trace_summary.py $t 0.05 > gpurun_out/r04_stats_$w/by_launch.txt
done
PVI_ROUND=r04 bash tools/collect_profiles.sh
ls profiles | grep r04
