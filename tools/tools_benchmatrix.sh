#!/bin/bash
# one bench line per BASELINE workload (GPU legs only except for the default line) -> gpurun_out/bench_lines.jsonl
OUT=gpurun_out/bench_lines.jsonl; : > $OUT
python bench.py 2>/dev/null | tail -1 >> $OUT
for w in c1 c2p c3 c4 c5; do python bench.py --workload $w --no-cpu 2>/dev/null | tail -1 >> $OUT; done
python - <<'PY'
import json
for l in open('gpurun_out/bench_lines.jsonl'):
    d = json.loads(l)
    print(d['config']['workload'][:44].ljust(44), 'ms/step %.4f' % d['ms_per_step'], ' cells/s %.3e' % d['value'], ' sweeps/s %.1f' % d['sweeps_per_sec'], d['dtype'])
PY
