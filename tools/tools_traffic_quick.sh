#!/bin/bash
# usage: tools_traffic_quick.sh <workload> [KEY=VALUE ...] -> HBM-side bytes per launch of the production sweep kernel
# (two rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE; the gfx950 correction of the guide: fetch x 2), for experiments with
# launch orders / bands.  Prints one line.
W=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/tq_$W; rm -rf $OUT; mkdir -p $OUT
for P in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/$P -o p -- python /root/repo/tools/tools_traffic.py $W "$@" > $OUT/$P.log 2>&1
done
python3 - "$W" "$*" <<PY
import csv, glob, sys, collections
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % c)
    rows = list(csv.DictReader(open(f[0])))
    geom = lambda r: (r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"))
    sw = [r for r in rows if "k_sweep" in r["Kernel_Name"] and "finish" not in r["Kernel_Name"] and "_probe" not in r["Kernel_Name"]]
    last = geom(sw[-1])
    v = [float(r["Counter_Value"]) for r in sw if geom(r) == last and r["Counter_Name"] == c]
    tot[c] = sum(v) / len(v)
print("TRAFFIC %s %s  fetch %.3f GB (x2 corrected)  write %.3f GB  total %.3f GB" % (sys.argv[1], sys.argv[2], 2 * tot["FETCH_SIZE"] * 1024 / 1e9, tot["WRITE_SIZE"] * 1024 / 1e9, (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / 1e9))
PY
