#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_perm.log; : > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree or cartpole" > gpurun_out/r04_perm_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_perm_tests.log | tail -5 >> $L
for a in "PERM=0" "PERM=1"; do
  for w in "c3 200" "c4 40"; do
    echo "== $w $a" >> $L
    timeout 300 python tools/tools_time.py $w $a 2>&1 | grep -E "TIME|nodes|rror" | cut -c1-400 >> $L
  done
done
bash tools/tools_ldsconf.sh c3 c3_perm0 PERM=0 TV0=19 TV1=51 >> $L 2>&1
bash tools/tools_ldsconf.sh c3 c3_perm1 PERM=1 TV0=19 TV1=51 >> $L 2>&1
bash tools/tools_ldsconf.sh c4 c4_perm0 PERM=0 TV0=55 TV1=26 >> $L 2>&1
bash tools/tools_ldsconf.sh c4 c4_perm1 PERM=1 TV0=55 TV1=26 >> $L 2>&1
cat $L
