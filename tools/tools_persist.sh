#!/bin/bash
cd /root/repo
for D in 0 1 16 17 32 33; do PVI_DBG=$D timeout 60 python tools/tools_ablate.py c2 300; done
for D in 0 1; do PVI_TV0=10 PVI_TV1=51 PVI_DBG=$D timeout 60 python tools/tools_ablate.py c3 10; done
