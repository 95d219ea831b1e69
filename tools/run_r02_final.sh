#!/bin/bash
# end-of-round pass: whole GPU test suite, profiles of every BASELINE workload, default bench under rocprof stats
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02_gputests_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_gputests_final.log
tail -14 gpurun_out/r02_gputests_final.log
bash tools/tools_profile_r02.sh 2>&1 | tail -12
