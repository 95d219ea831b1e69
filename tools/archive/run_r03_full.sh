#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r03_gputests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_gputests.log
tail -15 gpurun_out/r03_gputests.log
bash tools/tools_profile_r03.sh 2>&1 | tail -25
