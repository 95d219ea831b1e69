#!/bin/bash
# copies the round's summaries that tools_profile_<round>.sh left under gpurun_out/ into profiles/ (tracked) and rebuilds
# profiles/counters.json with the commit they were taken at
cd /root/repo
R=${PVI_ROUND:-r04}
export PVI_ROUND=$R
WL=${WL:-"c3 c4 c2 c2p c5 c5d c1 h3"}
python tools/make_counters_json.py $WL > /dev/null
for w in $WL default; do
  f=$(find gpurun_out/${R}_stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/${R}_${w}_bench_kernel_stats.csv
  [ -f gpurun_out/${R}_stats_$w/by_launch.txt ] && cp gpurun_out/${R}_stats_$w/by_launch.txt profiles/${R}_${w}_bench_by_launch.txt
done
f=$(find gpurun_out/${R}_stats_c3fb -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp $f profiles/${R}_c3fb_bench_kernel_stats.csv
cp gpurun_out/${R}_bench_c3fb.json profiles/${R}_bench_c3fb.json 2>/dev/null
cp gpurun_out/${R}_bench_default.json profiles/${R}_bench_default.json 2>/dev/null
cp gpurun_out/${R}_gputests.log profiles/${R}_gputests.log 2>/dev/null
cp gpurun_out/bench_full.json profiles/${R}_bench_default_full.json 2>/dev/null
true
