#!/bin/bash
# round 5, ONE call for a GPU window of unknown length: (1) the code written while the boxes were closed, safest first
# (tools/run_r05_validate.sh), (2) the whole GPU suite, (3) the round's profiles and counters (tools/tools_profile.sh).
# Every stage under its own timeout; the summaries land in gpurun_out/ as each stage ends.
cd /root/repo; mkdir -p gpurun_out
bash tools/run_r05_validate.sh > gpurun_out/r05_validate.out 2>&1; tail -8 gpurun_out/r05_validate.out
timeout 1500 python -m pytest tests -m gpu -q -k "not 2d_float32" > gpurun_out/r05_gputests.log 2>&1; echo "suite rc=$? :: $(tail -1 gpurun_out/r05_gputests.log)"
PVI_ROUND=r05 timeout 2400 bash tools/tools_profile.sh > gpurun_out/r05_profile.out 2>&1; tail -5 gpurun_out/r05_profile.out
timeout 900 bash tools/run_writecal.sh > gpurun_out/r05_writecal.out 2>&1; tail -12 gpurun_out/r05_writecal.out
# last, because it is the one piece that could leave the GPU spinning if it were wrong: the cooperative multi-sweep launch of the
# 2-D float32 sweep (opt-in, written blind) against one launch per sweep
timeout 300 python -m pytest tests/test_gpu_stress.py -m gpu -q -x -k "2d_float32" > gpurun_out/r05_multi32.log 2>&1; echo "multi32 rc=$? :: $(tail -1 gpurun_out/r05_multi32.log)"
