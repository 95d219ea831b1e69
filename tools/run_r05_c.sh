#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; L=gpurun_out/r05_c.log; : > $L
for rep in 1 2 3; do
 for w in c3 c4; do
  n=200; [ $w = c4 ] && n=40
  E="TV0=15 TV1=34"; [ $w = c4 ] && E="TV0=13 TV1=38"      # (an even tile width: RS_CONG applies)
  for ov in "" "VMASK=0" "ORDER=swapped" "$E" "ORDER=swapped $E" "ORDER=swapped $E RS_CONG=1"; do
   echo "== $w [$ov]" >> $L
   timeout 300 python tools/tools_time.py $w $n $ov 2>&1 | grep -E "TIME" | cut -c1-120 >> $L
  done
  echo "== $w [ablate-uniform]" >> $L
  PYROVI_LIB=/root/repo/pyro_amd/libpyrovi_ablate.so timeout 300 python tools/tools_time.py $w $n 2>&1 | grep -E "TIME" | cut -c1-120 >> $L
  echo "== $w [base]" >> $L
  PYROVI_LIB=/root/repo/pyro_amd/libpyrovi_base.so timeout 300 python tools/tools_time.py $w $n 2>&1 | grep -E "TIME" | cut -c1-120 >> $L
 done
done
paste - - < $L | awk '{print $2,$3,$(NF-1)}'
# the cart-pole with its coordinates swapped (lanes along the axis the displacement does not depend on): tools/tools_swapped_cartpole.py
S=gpurun_out/r05_swapped.log; : > $S
timeout 300 python tools/tools_swapped_cartpole.py 31 21 20 check 2>&1 | grep -E "TIME|J after|nodes" | cut -c1-400 >> $S
for rep in 1 2; do
  timeout 600 python tools/tools_swapped_cartpole.py 101 21 200 2>&1 | grep -E "TIME|nodes" | cut -c1-400 >> $S
  # (both orders with one pinned tiling of even width, the pitch as today and congruent to the width modulo 32)
  timeout 600 python tools/tools_swapped_cartpole.py 101 21 200 TV0=15 TV1=34 2>&1 | grep -E "TIME|nodes" | cut -c1-400 >> $S
  timeout 600 python tools/tools_swapped_cartpole.py 101 21 200 TV0=15 TV1=34 RS_CONG=1 2>&1 | grep -E "TIME|nodes" | cut -c1-400 >> $S
done
grep -E "TIME|J after" $S
