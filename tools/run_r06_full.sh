#!/bin/bash
# round 6, ONE GPU call (usage: bash tools/run_r06_full.sh <commit>): validate the library, record which kernels ran, then -- only
# if the suite is green -- the round's profiles, counters and default bench.  Every stage under its own timeout, summaries in
# gpurun_out/r06/ as each stage ends; tools/collect_r06.sh (run in the repository afterwards) copies them into profiles/.
C=${1:-unknown}
cd /root/repo; O=gpurun_out/r06; mkdir -p $O; : > $O/summary.log
say() { echo "$@" | tee -a $O/summary.log; }
t() { local n=$1 to=$2; shift 2
  timeout $to python -m pytest "$@" -m gpu -q > $O/$n.log 2>&1; local rc=$?; say "$n rc=$rc :: $(tail -1 $O/$n.log)"; return $rc; }
say "commit $C  $(date -u +%FT%TZ)  $(python tools/kernel_manifest.py check pyro_amd/kernel_manifest.json | tail -1)"
# 1. safest first: determinism / lockstep stress, then the whole suite in the driver's order (-x as the driver runs it) under a
#    kernel trace, so that the list of kernels that really executed comes back with the result
t stress 600 tests/test_gpu_stress.py || { say "STRESS FAILED: stopping"; exit 1; }
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q -x --durations=8 > $O/suite.log 2>&1; RC=$?
say "suite rc=$RC :: $(grep -E 'passed|failed|error' $O/suite.log | tail -1)"
if [ $RC -eq 0 ]; then
  # ... once more under a kernel trace: the list of kernels that really executed (children included: one stats file per process)
  (cd /tmp && export TMPDIR=/tmp && timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/suite_trace -- \
     python -m pytest /root/repo/tests/test_gpu_parity.py /root/repo/tests/test_gpu_stress.py -m gpu -q -x -p no:cacheprovider > /root/repo/$O/suite_traced.log 2>&1)
  say "suite under the tracer rc=$? :: $(grep -E 'passed|failed|error' $O/suite_traced.log | tail -1)"
  python - $O/suite_trace > $O/kernels_seen.txt <<'PY'
import csv, glob, os, sys
seen = set()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        seen.add(r.get("Name") or r.get("Kernel_Name") or "")
print("\n".join(sorted(k for k in seen if k)))
PY
  say "kernels seen by the suite: $(wc -l < $O/kernels_seen.txt)"
  ONLY=""; [ $(wc -l < $O/kernels_seen.txt) -gt 50 ] && ONLY="--only $O/kernels_seen.txt"
  python tools/kernel_manifest.py bless pyro_amd/kernel_manifest.json $ONLY --commit $C --file $O/verified_kernels.json \
    --evidence "round 6 GPU run on this commit's library: tests/test_gpu_parity.py + tests/test_gpu_stress.py green (gpurun_out/r06/suite.log)${ONLY:+; the kernels listed are the ones a second run under rocprofv3 --kernel-trace saw execute}" | tee -a $O/summary.log
fi
# 2. code that had never run (collected last by the driver too): each group on its own, no -x
t zz_swapped 600 tests/test_gpu_zz_unproven.py -k swapped
t zz_fb2d 900 tests/test_gpu_zz_unproven.py -k "feedback_storage_on_2d"
t zz_fbexp 600 tests/test_gpu_zz_unproven.py -k "explicit_system or node_table_tier"
t zz_rest 900 tests/test_gpu_zz_unproven.py -k "declared_invariance or refusals or python_driven or closed_loop or cubic or slinear or corruption_detector or whole_number"
[ $RC -eq 0 ] || { say "SUITE NOT GREEN: no profiles"; exit 1; }
# 3. the round's evidence: per-workload kernel-trace stats + four PMC passes, counters.json keyed on the ISA hash, default bench
PVI_ROUND=r06 timeout 2700 bash tools/tools_profile.sh > $O/profile.out 2>&1; say "profiles: $(tail -1 $O/profile.out | cut -c1-200)"
timeout 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err; say "bench rc=$? :: $(tail -c 300 $O/bench_final.json)"
timeout 600 bash tools/run_writecal.sh > $O/writecal.out 2>&1; say "writecal: $(tail -3 $O/writecal.out | tr '\n' ' ' | cut -c1-300)"
timeout 600 python tools/slab_time.py c4 8 0 3 > $O/slab_times.log 2>&1; say "slabs: $(tail -2 $O/slab_times.log | tr '\n' ' ' | cut -c1-300)"
# 4. last: the one kernel that could leave a GPU spinning if it were wrong (its barrier gives up after a second)
t zz_multi32 300 tests/test_gpu_zz_unproven.py -k "2d_float32" -rs && { timeout 300 python tools/time_multi32.py > $O/time_multi32.log 2>&1; say "multi32 timing: $(grep -c us/sweep $O/time_multi32.log) lines"; }
cat $O/summary.log
