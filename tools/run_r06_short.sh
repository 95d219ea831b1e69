#!/bin/bash
# round 6, the SHORT GPU call for a box that opens late (usage: bash tools/run_r06_short.sh <commit> [minutes, default 25]):
# what the round lacks most first -- smoke, the default bench line, the kernel-trace stats and the counter passes of the
# headline workload -- then parity: stress, the suite with -x as the driver runs it, and the never-run tests
# last.  Every stage under its own timeout, cut to what is left of the budget; summaries land in
# gpurun_out/r06/ as each stage ends (tools/collect_r06.sh copies them into profiles/).  tools/run_r06_full.sh is the long form.
C=${1:-unknown}; BUDGET=$(( ${2:-25} * 60 )); T0=$(date +%s)
cd /root/repo; O=gpurun_out/r06; mkdir -p $O; : > $O/summary.log
say() { echo "$@" | tee -a $O/summary.log; }
left() { local l=$(( BUDGET - ($(date +%s) - T0) )); [ $l -lt 0 ] && l=0; echo $l; }
# a stage's timeout: what it wants, at most what is left -- never 0 (`timeout 0` means NO limit): a spent budget gives every later stage one second
cap() { local want=$1 l=$(left); [ $want -gt $l ] && want=$l; [ $want -lt 1 ] && want=1; echo $want; }
say "commit $C  $(date -u +%FT%TZ)  budget ${BUDGET}s  $(python tools/kernel_manifest.py check pyro_amd/kernel_manifest.json | tail -1)"
timeout $(cap 300) python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; say "smoke rc=$? :: $(tail -1 $O/smoke.log | cut -c1-200)"
timeout $(cap 400) python bench.py > $O/bench_final.json 2> $O/bench_final.err; say "bench rc=$? :: $(tail -c 400 $O/bench_final.json)"
(cd /tmp && export TMPDIR=/tmp && timeout $(cap 300) rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r06_stats_default -o s -- \
   python /root/repo/bench.py --no-cpu > /root/repo/gpurun_out/r06_bench_default.json 2> /root/repo/gpurun_out/r06_bench_default.err)
say "bench under rocprofv3 --kernel-trace --stats rc=$? :: $(tail -c 300 gpurun_out/r06_bench_default.json)"
t=$(find gpurun_out/r06_stats_default -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/kernel_trace_summary.py $t 0.05 > gpurun_out/r06_stats_default/by_launch.txt
PVI_ROUND=r06 timeout $(cap 420) bash tools/tools_counters.sh c3 > gpurun_out/r06_counters_c3.log 2>&1; say "counter passes c3 rc=$? :: $(tail -1 gpurun_out/r06_counters_c3.log | cut -c1-200)"
timeout $(cap 300) python -m pytest tests/test_gpu_stress.py -m gpu -q > $O/stress.log 2>&1; RS=$?; say "stress rc=$RS :: $(tail -1 $O/stress.log)"
timeout $(cap 1500) python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q -x --durations=8 > $O/suite.log 2>&1; RC=$?
say "suite rc=$RC :: $(grep -E 'passed|failed|error' $O/suite.log | tail -1)"
# (no bless here: blessing needs the list of kernels that really executed -- the traced second run of tools/run_r06_full.sh)
[ $RC -eq 0 ] && PVI_ROUND=r06 PVI_HEAD=$C python tools/make_counters_json.py c3 >> $O/summary.log 2>&1
for g in "swapped" "feedback_storage_on_2d" "explicit_system or node_table_tier" "declared_invariance or refusals or python_driven or closed_loop or cubic or slinear or corruption_detector or whole_number" "2d_float32"; do
  n=zz_$(echo "$g" | cut -d' ' -f1); l=$(cap 420); [ $l -lt 30 ] && { say "$n: no time left"; continue; }
  timeout $l python -m pytest tests/test_gpu_zz_unproven.py -m gpu -q -k "$g" > $O/$n.log 2>&1; say "$n rc=$? :: $(tail -1 $O/$n.log)"
done
cat $O/summary.log
