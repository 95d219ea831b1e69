#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r03_shapes.log; : > $L
python tools/tools_time.py c3 40 >> $L 2>&1
for s in "10 51" "5 51" "10 34" "15 34" "7 34" "20 25" "10 26" "19 26" "10 17" "20 17"; do set -- $s
  python tools/tools_time.py c3 40 TV0=$1 TV1=$2 | grep TIME >> $L 2>&1
done
python tools/tools_time.py c3 40 TV0=10 TV1=34 NO_XCD=1 | grep TIME >> $L 2>&1
python tools/tools_time.py c3 40 TV0=10 TV1=34 TABLES=0 | grep TIME >> $L 2>&1
python tools/tools_time.py c3 40 TV0=10 TV1=34 LDS_KB=160 | grep TIME >> $L 2>&1
python tools/tools_time.py c3 40 TV0=20 TV1=25 LDS_KB=160 | grep TIME >> $L 2>&1
cat $L
bash tools/tools_counters.sh c3 > gpurun_out/r03_counters_c3_first.log 2>&1; tail -3 gpurun_out/r03_counters_c3_first.log
