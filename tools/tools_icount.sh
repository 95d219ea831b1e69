#!/bin/bash
# instruction counts per wave of the sweep kernel for a workload spec (one rocprofv3 --pmc pass, kernel-trace only)
# usage: tools_icount.sh <workload> [<workload> ...]
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  T=$(echo $W | tr ':,' '__')
  OUT=/root/repo/gpurun_out/icount_$T; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -o p -- python /root/repo/tools/tools_traffic.py $W LSPLIT=0 > $OUT.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'sweep' not in k: continue
    w = sum(d['SQ_WAVES']) / len(d['SQ_WAVES'])
    print('$W', k, 'waves %d' % w, ' '.join('%s %.0f' % (c[8:], sum(v) / len(v) / w) for c, v in sorted(d.items()) if c != 'SQ_WAVES'))
PY
done
