#!/usr/bin/env python3
"""usage: tools_time.py <workload> [sweeps] [KEY=VALUE ...]   -- ms per sweep of one workload with variant overrides
(pvi_override keys, e.g. WIN=0 TV0=10 TV1=51 NO_XCD=1; ORDER=swapped: DynamicProgramming(internal_order="swapped")); prints the kernel
path.  For profiling passes and shape sweeps."""
import contextlib, io, sys, time
sys.path.insert(0, "/root/repo")
from pyro_amd import _native, configs
from pyro_amd.planning import dynamicprogramming
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 and "=" not in sys.argv[2] else 20
ov = dict(a.split("=", 1) for a in sys.argv[2:] if "=" in a)
order = ov.pop("ORDER", None)
with _native.overrides(**ov):
    cfg = configs.build(name)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=cfg["dtype"], internal_order=order)
    p = dp._p
    p.synchronize()
    t1 = time.time()
    p.sweep(3, 1.0, -1.0)
    p.sweep(n, 1.0, -1.0)
    print("nodes %d %s" % (cfg["grid_sys"].nodes_n, p.describe()))
    if order:
        ov["ORDER"] = order
    print("TIME %s %s setup %.2f s  %.4f ms/sweep" % (name, ov, t1 - t0, p.last_sweep_ms() / n), flush=True)
    J = p.get_J()
    p.close()
