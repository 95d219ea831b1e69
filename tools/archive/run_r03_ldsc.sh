#!/bin/bash
# LDS counters of the lean4 kernel for a few forced tile shapes
cd /tmp && export TMPDIR=/tmp
for s in "7 34" "11 34" "10 51" "10 17" "8 32"; do set -- $s
  OUT=/root/repo/gpurun_out/ldsc_$1x$2; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- python /root/repo/tools/tools_time.py c3 6 TV0=$1 TV1=$2 TV_EXACT=1 > $OUT/log 2>&1
  python3 - <<PY
import csv, glob, collections
fs = glob.glob('$OUT/*counter_collection.csv')
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'k_sweep_lean4' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v)/len(v) for k, v in acc.items()}
print('$1x$2', {k: round(v/1e6,1) for k,v in m.items()}, 'cyc/LDSinstr %.2f' % (m['SQ_LDS_IDX_ACTIVE']/m['SQ_INSTS_LDS']), 'conflict share %.2f' % (m['SQ_LDS_BANK_CONFLICT']/m['SQ_LDS_IDX_ACTIVE']), 'lds busy %.2f' % (m['SQ_LDS_IDX_ACTIVE']/256/(m['GRBM_GUI_ACTIVE']/8)))
PY
  grep TIME $OUT/log
done
