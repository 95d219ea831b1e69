"""Set-up cost and stability of the timed tile choice: every workload is created several times in fresh PROCESSES' worth of
state (TUNE=2 ignores the per-process cache) and once more from the cache; prints create seconds, the chosen tile and
the sweep time.  usage: tools_create_time.py [workload ...] [repeats=N]"""
import sys, time, io, contextlib
sys.path.insert(0, "/root/repo")
from pyro_amd import _native, configs
from pyro_amd.planning import dynamicprogramming
names = [a for a in sys.argv[1:] if "=" not in a] or ["c3", "c4"]
rep = int(dict(a.split("=") for a in sys.argv[1:] if "=" in a).get("repeats", 5))
for name in names:
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
    for i in range(rep + 1):
        ov = {"TUNE": "2"} if i < rep else {}
        with _native.overrides(**ov), contextlib.redirect_stdout(io.StringIO()):
            t0 = time.time()
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=cfg["dtype"])
            dp._p.synchronize()
            t1 = time.time() - t0
        p = dp._p
        n = 20 if name in ("c3",) else 8
        p.sweep(2, 1.0, -1.0)
        p.sweep(n, 1.0, -1.0)
        d = dict(t.split("=", 1) for t in p.describe().split() if "=" in t)
        print("CREATE %s %s create %.2f s  tile=%s block=%s lds=%s  %.3f ms/sweep  cands=%s" % (
            name, "timed " if i < rep else "cached", t1, d.get("tile"), d.get("block"), d.get("lds_bytes"), p.last_sweep_ms() / n,
            d.get("cands", "")), flush=True)
        p.close()
