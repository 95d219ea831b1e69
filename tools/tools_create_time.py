import sys, time, io, contextlib
sys.path.insert(0, "/root/repo")
from pyro_amd import configs
from pyro_amd.planning import dynamicprogramming
for name in ("c3", "c4", "c2"):
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
        t0 = time.time()
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=cfg["dtype"])
        dp._p.synchronize()
        t1 = time.time() - t0
    print(name, "create %.2f s" % t1, dp._p.describe()[:40])
    dp._p.close()
