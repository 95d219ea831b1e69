#!/bin/bash
# round 5, fault hunt 3: the e1 build's error-feedback kernel returns NaN from the first sweeps (hunt 1) -- which nodes, and
# which of the two source changes does it?
cd /root/repo; mkdir -p gpurun_out; O=gpurun_out/r05_hunt3; mkdir -p $O
export PYTHONUNBUFFERED=1
for t in e0 e1 e1a e1b; do
  lib=libpyrovi_$t.so; [ $t = e0 ] && lib=libpyrovi.so
  PYROVI_LIB=/root/repo/pyro_amd/$lib timeout 120 python tools/r05_hunt/hunt_fb.py $t > $O/fb_$t.log 2>&1; echo "fb $t rc=$?" >> $O/fb_$t.log
  PYROVI_LIB=/root/repo/pyro_amd/$lib timeout 120 python tools/r05_hunt/hunt_fb.py s$t --cfg cartpole:41,41,41,41:21:float32 > $O/fbs_$t.log 2>&1; echo "fb small $t rc=$?" >> $O/fbs_$t.log
done
for t in e1 e1a e1b; do
  python tools/r05_hunt/hunt_cmp.py e0 $t 101,101,101,101 > $O/cmp_$t.log 2>&1
  python tools/r05_hunt/hunt_cmp.py se0 s$t 41,41,41,41 > $O/cmps_$t.log 2>&1
done
tail -n 40 $O/*.log
