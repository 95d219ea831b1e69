#!/bin/bash
# usage: tools_ldsconf.sh <workload> <tag>  (env selects the configuration); prints LDS counters of the sweep kernel
W=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/lds_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT -o p -- python /root/repo/bench.py --workload $W --no-cpu --steps 3 --warmup 1 > $OUT/log 2>&1
python3 - <<PY
import csv, glob, collections
fs = glob.glob('$OUT/*counter_collection.csv')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    acc[r['Kernel_Name'][:24]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'sweep' in k: print('$TAG', {c: '%.3g' % (sum(v)/len(v)) for c, v in d.items()})
PY
grep -o "path=[^\"]*" $OUT/log | head -1 | cut -c1-120
