#!/bin/bash
# round-2 profiles: for every BASELINE workload (a) four rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | two SQ sets,
# kernel-trace only) -> gpurun_out/counters_<w>.json, (b) kernel-trace stats of `bench.py --workload <w> --no-cpu`;
# plus kernel-trace stats of the default `python bench.py`.  4-D tile shapes are pinned to what set-up timing picks, so
# that the tuning sweeps of other shapes stay out of the per-kernel averages.
cd /root/repo
export PVI_ROUND=r02 PVI_HEAD=${PVI_HEAD:-unknown}
WL=${WL:-"c3 c2 c2p c4 c5 c1"}
for w in $WL; do
  unset PVI_TV0 PVI_TV1 PVI_TV_EXACT
  if [ $w = c3 ] || [ $w = c4 ]; then
    SHAPE=$(python tools/tools_describe.py $w | grep -o 'tile=[0-9]*x[0-9]*' | head -1)
    export PVI_TV0=$(echo $SHAPE | sed 's/tile=\([0-9]*\)x.*/\1/') PVI_TV1=$(echo $SHAPE | sed 's/.*x//') PVI_TV_EXACT=1
    echo "$w shape $SHAPE" >> gpurun_out/r02_shapes.log
  fi
  bash tools/tools_counters.sh $w > gpurun_out/r02_counters_$w.log 2>&1
  S=""; ([ $w = c3 ] || [ $w = c4 ] || [ $w = c5 ]) && S="--steps 10 --warmup 2"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02_stats_$w -o s -- python /root/repo/bench.py --workload $w --no-cpu $S > /root/repo/gpurun_out/r02_stats_$w.log 2>&1)
  tail -1 gpurun_out/r02_stats_$w.log | cut -c1-200
done
unset PVI_TV0 PVI_TV1 PVI_TV_EXACT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02_stats_default -o s -- python /root/repo/bench.py > /root/repo/gpurun_out/r02_bench_default.json 2> /root/repo/gpurun_out/r02_bench_default.err)
tail -c 600 gpurun_out/r02_bench_default.json
find gpurun_out -name "*kernel_stats.csv" | head -20
