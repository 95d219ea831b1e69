#!/bin/bash
# usage: tools_counters.sh <workload> -> gpurun_out/counters_<workload>.json
# separate rocprofv3 --pmc passes (kernel-trace only): FETCH_SIZE | WRITE_SIZE | SQ set 1 | SQ set 2
W=$1
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/counters_$W; mkdir -p $OUT
# lean sweeps (2-D and 4-D): the timed choice between near-equal tile shapes flips under the profiler (counter collection serialises
# the launches): take the shape an UNPROFILED create chooses and pin it for the passes, so that the counters describe the
# variant bench.py runs
PIN=$(python /root/repo/tools/tools_describe.py $W 2>/dev/null | python3 -c "
import sys,re
d=dict(t.split('=',1) for t in sys.stdin.read().split() if '=' in t)
if d.get('path')=='lean' and d.get('win')=='0' and d.get('lsplit')=='0' and 'tile' in d:
    a,b=d['tile'].split('x'); print('TV0=%s TV1=%s NPT=%s' % (a,b,d.get('npt','1')))
elif d.get('path')=='lean' and d.get('win')=='1' and d.get('choice','-')!='-':
    print('L4PIN=%s' % d['choice'])
")
echo "pins: $PIN"
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o p -- python /root/repo/tools/tools_traffic.py $W $PIN > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for i in (1, 2, 3, 4):
    fs = glob.glob('$OUT/p%d/*counter_collection.csv' % i)
    if not fs: print('no counters pass', i); continue
    # pvi_create times candidate variants (tile shapes, mappings) with real sweeps of the SAME kernel name: keep, per
    # kernel, only the dispatches that have the launch geometry of its LAST dispatch -- the production launches
    rows = list(csv.DictReader(open(fs[0])))
    geom = lambda r: (r.get('Grid_Size'), r.get('Workgroup_Size'), r.get('LDS_Block_Size'))
    last = {}
    for r in rows:
        last[r['Kernel_Name'].split('(')[0][:96]] = geom(r)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r['Kernel_Name'].split('(')[0][:96]
        if geom(r) == last[k]:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        for c, v in d.items():
            res[k][c] = sum(v) / len(v)
            res[k]['calls'] = len(v)
keep = {k: v for k, v in res.items() if 'sweep' in k or 'to_f64' in k or 'setup' in k}
path = ''
try:
    path = [l for l in open('$OUT/p1.log') if l.startswith('nodes ')][-1].split(' ', 2)[2].strip()
except Exception as e:
    print('no kernel path in the log:', e)
import sys
sys.path.insert(0, '/root/repo')
import bench
# (the ISA hash of every kernel counted, from the manifest of the library the passes ran on: bench.py drops the counters once a
#  kernel is compiled to other instructions)
isa = {bench.norm_kernel(k): bench.kernel_isa_hash(k) for k in keep}
json.dump({"workload": "$W", "kernel_path": path, "isa_hashes": isa, "kernels": keep}, open('/root/repo/gpurun_out/counters_$W.json', 'w'), indent=1)
for k, v in keep.items():
    if 'sweep' in k: print(k, json.dumps(v))
PY
