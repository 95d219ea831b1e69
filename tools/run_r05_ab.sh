#!/bin/bash
# A/B on one box: usage run_r05_ab.sh <tag> <libA> <libB> [workloads...]   (three alternating runs each)
cd /root/repo; mkdir -p gpurun_out; tag=$1; A=$2; B=$3; shift 3; W=${@:-c3 c4}
L=gpurun_out/r05_ab_$tag.log; : > $L
for rep in 1 2 3; do
for lib in $A $B; do
  for w in $W; do
  n=200; [ $w = c4 ] && n=40; [ $w = c5 ] && n=40; [ $w = c2 ] && n=4000; [ $w = c2p ] && n=4000; [ $w = c1 ] && n=4000
  echo "== $lib $w" >> $L
  PYROVI_LIB=/root/repo/pyro_amd/$lib timeout 300 python tools/tools_time.py $w $n 2>&1 | grep -E "TIME|rror" | cut -c1-200 >> $L
  done
done
done
cat $L
