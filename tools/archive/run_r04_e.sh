#!/bin/bash
# late round 4: autotune after the cold-start fix (fresh process per create), the new tests, the feedback kernel's timing
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_autotune_after.log; : > $L
for i in 1 2 3 4 5 6; do timeout 300 python tools/tools_time.py c3 200 2>&1 | grep -E "TIME|nodes|rror" | sed -e "s/.*choice=/choice=/" | cut -c1-700 >> $L; done
for i in 1 2 3; do timeout 300 python tools/tools_time.py c4 40 2>&1 | grep -E "TIME|nodes|rror" | sed -e "s/.*choice=/choice=/" | cut -c1-900 >> $L; done
timeout 300 python tools/tools_time.py cartpole:41,41,41,41:21:float32 500 2>&1 | grep -E "TIME|rror" | cut -c1-700 >> $L
cat $L | cut -c1-330
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "feedback or variants_agree or caller_transport or world1 or config1" > gpurun_out/r04_e_tests.log 2>&1; tail -4 gpurun_out/r04_e_tests.log
bash tools/run_r04_d.sh 2>&1 | tail -4
