#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_regtab_wide.log; : > $L
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_sweep or config1 or golden or pendulum or c1" > gpurun_out/r04_regtab_wide_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_regtab_wide_tests.log | tail -8 >> $L
timeout 900 python tools/tools_fuzz64.py 120 31 > gpurun_out/r04_fuzz64c.log 2>&1; tail -1 gpurun_out/r04_fuzz64c.log >> $L; grep -c "regtab=1" gpurun_out/r04_fuzz64c.log >> $L
for lib in libpyrovi_prev.so libpyrovi.so; do
  echo "== $lib c1" >> $L
  PYROVI_LIB=/root/repo/pyro_amd/$lib timeout 300 python tools/tools_time.py c1 3000 2>&1 | grep -E "TIME|rror" | cut -c1-200 >> $L
done
for w in "pendulum:201,201:21:float64" "pendulum:201,201:11:float64" "pendulum:301,301:21:float64" "pendulum:101,101:21:float64" "pendulum:51,401:24:float64" "pendulum:401,51:13:float64" "pendulum:301,301:11:float64"; do
for a in "MULTI=0" ""; do
  echo "== $w $a" >> $L
  timeout 300 python tools/tools_time.py $w 2000 $a 2>&1 | grep -E "TIME|rror|nodes" | cut -c1-260 >> $L
done
done
cat $L
