#!/bin/bash
# kernel-trace stats + HBM counters (separate passes) of the packed table-tier sweep (pendulum 1001^2 x 51, float32)
W=pendulum:1001,1001:51:float32
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/table_prof; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python /root/repo/tools/tools_table.py $W > $OUT/stats.log 2>&1
for P in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/$P -o p -- python /root/repo/tools/tools_table.py $W > $OUT/$P.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
res = {}
for r in csv.DictReader(open('$OUT/stats/s_kernel_stats.csv')):
    if 'tablep' in r['Name']: res['avg_ns'] = float(r['AverageNs']); res['calls'] = int(r['Calls'])
for P in ('FETCH_SIZE', 'WRITE_SIZE'):
    v = [float(r['Counter_Value']) for f in glob.glob('$OUT/%s/*counter_collection.csv' % P) for r in csv.DictReader(open(f)) if 'tablep' in r['Kernel_Name'] and r['Counter_Name'] == P]
    res[P + '_kib_raw'] = sum(v) / max(len(v), 1)
cells = 1001 * 1001 * 51
res['record_bytes'] = cells * 16
res['hbm_bytes_per_launch'] = (2.0 * res['FETCH_SIZE_kib_raw'] + res['WRITE_SIZE_kib_raw']) * 1024   # FETCH_SIZE reports 1/2 on gfx950
res['achieved_GBps_on_records'] = res['record_bytes'] / res['avg_ns']
res['achieved_GBps_on_counted_traffic'] = res['hbm_bytes_per_launch'] / res['avg_ns']
json.dump(res, open('/root/repo/gpurun_out/table_prof.json', 'w'), indent=1)
print(json.dumps(res))
PY
cp $OUT/stats/s_kernel_stats.csv /root/repo/gpurun_out/table_kernel_stats.csv
