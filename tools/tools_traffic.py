#!/usr/bin/env python3
"""Workload for the HBM-traffic counter passes: a few sweeps of <workload> plus one download of J
(k_to_f64: reads N*4 B, writes N*8 B -- a known byte count used to calibrate FETCH_SIZE / WRITE_SIZE)."""
import contextlib, io, sys
sys.path.insert(0, "/root/repo")
from pyro_amd import configs
from pyro_amd.planning import dynamicprogramming
from pyro_amd import _native
name = sys.argv[1]
ov = dict(a.split("=", 1) for a in sys.argv[2:] if "=" in a)     # pvi_override pins (the variant an unprofiled run chose)
cfg = configs.build(name)
with _native.overrides(**ov), contextlib.redirect_stdout(io.StringIO()):
    dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=cfg["dtype"])
dp.save_time_history = False
p = dp._p
for _ in range(6):          # one sweep per batch: a multi-sweep launch (k_sweep64m) then is ONE sweep per launch too
    p.sweep(1, 1.0, -1.0)
J = p.get_J()
print("nodes", cfg["grid_sys"].nodes_n, p.describe())
