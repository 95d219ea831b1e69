#!/bin/bash
# usage: tools_traffic.sh <workload> ; separate PMC passes (FETCH_SIZE, WRITE_SIZE), prints per-kernel averages
W=$1
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/traffic_$W; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o p -- python /root/repo/tools/tools_traffic.py $W > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob('$OUT/%s/*counter_collection.csv' % C)
    if not fs: print('no counters', C); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r['Counter_Name'] == C:
            acc[r['Kernel_Name'].split('(')[0][:60]].append(float(r['Counter_Value']))
    for k, v in acc.items():
        res[k][C] = sum(v) / len(v); res[k]['calls'] = len(v)
print(json.dumps({"workload": "$W", "kernels": res}))
json.dump({"workload": "$W", "kernels": res}, open('$OUT/summary.json', 'w'), indent=1)
PY
grep nodes $OUT/FETCH_SIZE.log
