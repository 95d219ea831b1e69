#!/bin/bash
# after tools/run_r06_full.sh (or tools/run_r06_short.sh) came back: copy the round's summaries from gpurun_out/ into profiles/ (tracked)
cd /root/repo; O=gpurun_out/r06
PVI_ROUND=r06 bash tools/collect_profiles.sh
for f in summary.log smoke.log suite.log stress.log zz_swapped.log zz_fb2d.log zz_fbexp.log zz_multi32.log zz_feedback_storage_on_2d.log zz_explicit_system.log zz_rest.log zz_declared_invariance.log zz_2d_float32.log kernels_seen.txt time_multi32.log slab_times.log bench_final.json; do
  [ -f $O/$f ] && cp $O/$f profiles/r06_$f
done
f=$(find $O/suite_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/r06_suite_kernel_stats.csv
[ -f gpurun_out/r06_writecal.log ] && cp gpurun_out/r06_writecal.log profiles/
[ -f $O/verified_kernels.json ] && cp $O/verified_kernels.json profiles/verified_kernels.json
ls profiles | grep r06
