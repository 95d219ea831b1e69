#!/bin/bash
./tools_b1.sh c3 5 1
PVI_RS_MODE=2 ./tools_b1.sh c3 5 1
PVI_TV0=12 PVI_TV1=51 PVI_LDS_KB=80 ./tools_b1.sh c3 5 1
./tools_b1.sh c4 3 1
./tools_b1.sh c2 200 20
