#!/bin/bash
cd /root/repo
for lib in libpyrovi_t2.so libpyrovi_t3.so libpyrovi.so; do
  echo "== $lib"
  PYROVI_LIB=/root/repo/pyro_amd/$lib WL="c1 pendulum:1001,1001:51:float64 pendulum:201,201:201:float64 pendulum:2001,2001:21:float64" bash tools/run_all_times.sh
done
