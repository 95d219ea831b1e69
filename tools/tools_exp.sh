#!/bin/bash
for sh in "16 26" "12 26" "4 51" "8 34" "10 34" "19 21"; do set -- $sh; PVI_TV0=$1 PVI_TV1=$2 PVI_LDS_KB=80 PVI_MARCH_LDS_KB=80 ./tools/tools_b1.sh c3 5 1; done
PVI_TV0=10 PVI_TV1=51 PVI_LDS_KB=80 PVI_MARCH_LDS_KB=150 ./tools/tools_b1.sh c3 5 1
PVI_TV0=8 PVI_TV1=51 PVI_LDS_KB=80 PVI_MARCH_LDS_KB=110 ./tools/tools_b1.sh c3 5 1
