#!/bin/bash
# round 5, fault hunt 12: scratch privacy at full occupancy (tools/r05_hunt/scratchtest.hip)
cd /root/repo; mkdir -p gpurun_out; O=gpurun_out/r05_hunt12; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/scratchtest tools/r05_hunt/scratchtest.hip > $O/build.log 2>&1
timeout 200 /tmp/scratchtest 24544 8528 40 > $O/scratchtest_small.log 2>&1; echo "rc=$?" >> $O/scratchtest_small.log
timeout 200 /tmp/scratchtest 43264 60000 60 > $O/scratchtest_c3.log 2>&1; echo "rc=$?" >> $O/scratchtest_c3.log
cat $O/scratchtest_*.log
