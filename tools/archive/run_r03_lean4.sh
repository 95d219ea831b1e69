#!/bin/bash
# round 3: the 4-D lean kernel -- variant agreement, full-size sampled parity, convergence drift, timings
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree or self_check" > gpurun_out/r03_lean4_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03_lean4_tests.log; tail -30 gpurun_out/r03_lean4_tests.log
python - <<'PY' 2>&1 | tee gpurun_out/r03_lean4_times.log
import contextlib, io, sys, time
sys.path.insert(0, "/root/repo")
from pyro_amd import configs, _native
from pyro_amd.planning import dynamicprogramming
for name, n in (("c3", 40), ("c4", 10)):
    for ov in ({}, {"WIN": 0}):
        with _native.overrides(**ov):
            cfg = configs.build(name)
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype="float32")
            p = dp._p
            p.synchronize(); t1 = time.time()
            p.sweep(5, 1.0, -1.0)
            p.sweep(n, 1.0, -1.0); ms = p.last_sweep_ms() / n
            print(name, ov, "setup %.2f s  %.3f ms/sweep" % (t1 - t0, ms), p.describe(), flush=True)
            p.close()
PY
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "sampled_against_c_oracle and (c3 or c4) or solved_to_tolerance" > gpurun_out/r03_lean4_full.log 2>&1
echo "rc=$?" >> gpurun_out/r03_lean4_full.log; grep -E "after|sweeps|passed|failed|Error|assert" gpurun_out/r03_lean4_full.log | tail -40
