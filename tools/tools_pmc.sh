#!/bin/bash
# usage: tools_pmc.sh <workload> <tag> [extra bench args]; runs counter passes, prints per-kernel averages
W=$1; TAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_$TAG; mkdir -p $OUT
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P3="SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o p -- python /root/repo/bench.py --workload $W --no-cpu --steps 5 --warmup 1 "$@" > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in (1,2,3):
    fs = glob.glob('$OUT/p%d/*counter_collection.csv' % i)
    if not fs: print('no counters pass', i); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        if 'sweep' not in k: continue
        print(k, {c: '%.4g' % (sum(v)/len(v)) for c, v in d.items()})
PY
