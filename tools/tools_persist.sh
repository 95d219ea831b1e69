#!/bin/bash
cd /root/repo
for S in "10 51" "5 51" "7 34" "15 34" "8 26" "10 26" "6 41" "12 41"; do set -- $S; PVI_TV0=$1 PVI_TV1=$2 PVI_TV_EXACT=1 timeout 60 python tools/tools_ablate.py c3 10 | cut -c1-150; done
