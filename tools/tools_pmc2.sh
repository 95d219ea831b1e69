#!/bin/bash
# usage: tools_pmc2.sh <workload> <tag> ; two PMC passes; env selects configuration
W=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc2_$TAG; mkdir -p $OUT
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o p -- python /root/repo/bench.py --workload $W --no-cpu --steps 3 --warmup 1 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = {}
for i in (1,2):
    fs = glob.glob('$OUT/p%d/*counter_collection.csv' % i)
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        acc[r['Kernel_Name'][:22]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        if 'sweep' in k:
            for c, v in d.items(): tot[c] = sum(v)/len(v)
w = tot.get('SQ_WAVES', 1)
print('$TAG waves %.3g  VALU/wave %.0f  LDSinst/wave %.0f  wavecyc(quad)/wave %.0f  active%% %.0f wait_any%% %.0f wait_inst%% %.0f | VALUbusy_ms %.2f LDSbusy_ms %.2f ldscyc/inst %.1f conf/inst %.1f GUI_ms %.2f' % (
  w, tot['SQ_INSTS_VALU']/w, tot['SQ_INSTS_LDS']/w, tot['SQ_WAVE_CYCLES']/w,
  100*tot['SQ_ACTIVE_INST_ANY']/tot['SQ_WAVE_CYCLES'], 100*tot['SQ_WAIT_ANY']/tot['SQ_WAVE_CYCLES'], 100*tot['SQ_WAIT_INST_ANY']/tot['SQ_WAVE_CYCLES'],
  tot['SQ_ACTIVE_INST_VALU']*4/1024/2.4e6, tot['SQ_LDS_IDX_ACTIVE']/256/2.4e6, tot['SQ_LDS_IDX_ACTIVE']/tot['SQ_INSTS_LDS'], tot['SQ_LDS_BANK_CONFLICT']/tot['SQ_INSTS_LDS'], tot['GRBM_GUI_ACTIVE']/8/2.4e6))
PY
