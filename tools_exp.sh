#!/bin/bash
for sh in "1 251" "2 251" "2 126" "4 126" "4 64" "8 64" "3 167" "8 32" "6 84"; do set -- $sh; PVI_TV0=$1 PVI_TV1=$2 ./tools_b1.sh c2 200 20; done
./tools_b1.sh c2p 200 20
for sh in "2 16" "4 16" "4 8" "8 8"; do set -- $sh; PVI_TV0=$1 PVI_TV1=$2 ./tools_b1.sh c2p 200 20; done
