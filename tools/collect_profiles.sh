#!/bin/bash
# copies the round-3 summaries that tools_profile_r03.sh left under gpurun_out/ into profiles/ (tracked) and rebuilds
# profiles/counters.json with the commit they were taken at
cd /root/repo
WL=${WL:-"c3 c4 c2 c2p c5 c5d c1 h3"}
python tools/make_counters_json.py $WL > /dev/null
for w in $WL default; do
  f=$(find gpurun_out/r03_stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/r03_${w}_bench_kernel_stats.csv
  [ -f gpurun_out/r03_stats_$w/by_launch.txt ] && cp gpurun_out/r03_stats_$w/by_launch.txt profiles/r03_${w}_bench_by_launch.txt
done
cp gpurun_out/r03_bench_default.json profiles/r03_bench_default.json 2>/dev/null
cp gpurun_out/r03_gputests.log profiles/r03_gputests.log 2>/dev/null
true
