#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_setup.log; : > $L
timeout 900 python tools/tools_create_time.py c4 repeats=6 2>&1 | grep -E "CREATE|rror" >> $L
cat $L
