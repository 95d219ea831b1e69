#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
L=gpurun_out/r04_xcd3.log; : > $L
for a in "" "XCD3=1"; do
  echo "== h3 $a" >> $L
  timeout 300 python tools/tools_time.py h3 500 $a 2>&1 | grep -E "TIME|rror" | cut -c1-300 >> $L
done
PVI_X= timeout 600 python - >> $L 2>&1 <<'PY'
import sys, numpy as np, contextlib, io
sys.path.insert(0, "/root/repo")
from pyro_amd import _native, configs
from pyro_amd.planning import dynamicprogramming
outs = {}
for ov in ({}, {"XCD3": "1"}):
    with _native.overrides(**ov):
        cfg = configs.build("h3")
        with contextlib.redirect_stdout(io.StringIO()):
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=cfg["dtype"])
        p = dp._p
        st, n = p.sweep(7, 1.0, -1.0)
        outs[str(ov)] = (p.get_J(), p.get_pi(), st)
        p.close()
a, b = outs.values()
print("identical J", np.array_equal(a[0], b[0]), "pi", np.array_equal(a[1], b[1]), "stats", np.array_equal(a[2], b[2]))
PY
cd /tmp && export TMPDIR=/tmp
for a in "" "XCD3=1"; do
  rm -rf /tmp/t3; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/t3 -o p -- python /root/repo/tools/tools_time.py h3 5 $a > /dev/null 2>&1
  python3 - "$a" >> /root/repo/$L <<'PY'
import csv, glob, sys, collections
fs = glob.glob('/tmp/t3/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'k_sweep3_fast' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('FETCH_SIZE raw (KiB?)', sys.argv[1], {k: sum(v)/len(v) for k, v in acc.items()})
PY
done
cat /root/repo/$L
