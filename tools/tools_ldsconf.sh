#!/bin/bash
# usage: tools_ldsconf.sh <workload> <tag> [KEY=VALUE ...]   -- LDS counters of the production sweep kernel under pvi_override keys
W=$1; TAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/lds_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- python /root/repo/tools/tools_time.py $W 4 "$@" > $OUT/log 2>&1
python3 - <<PY
import csv, glob, collections
fs = glob.glob('$OUT/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    acc[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'k_sweep' in k and 'probe' not in k and 'finish' not in k:
        m = {c: sum(v) / len(v) for c, v in d.items()}
        print('$TAG', k, 'launches', len(list(d.values())[0]), {c: '%.4g' % v for c, v in m.items()},
              'conflict/active %.2f  cycles/inst %.2f  lds busy %.2f' % (m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE'], m['SQ_LDS_IDX_ACTIVE'] / m['SQ_INSTS_LDS'],
                                                           m['SQ_LDS_IDX_ACTIVE'] / 256 / (m['GRBM_GUI_ACTIVE'] / 8)))
PY
grep -E "TIME" $OUT/log | cut -c1-160
