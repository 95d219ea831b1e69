import contextlib, io, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from pyro_amd import configs
from pyro_amd.planning import dynamicprogramming
name = sys.argv[1]; tol = float(sys.argv[2])
res = {}
for dt in ("float64", "float32"):
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=dt)
    dp.save_time_history = False; dp.verbose = False
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        dp.solve_bellman_equation(tol=tol)
    res[dt] = (dp.J.copy(), dp.pi.copy(), dp.k, time.time() - t0)
    print(dt, "sweeps", dp.k, "seconds %.2f" % res[dt][3], dp._p.describe()[:40], flush=True)
J64, J32 = res["float64"][0], res["float32"][0]
print("rel err max|dJ|/max|J| = %.3e ; pi mismatch %.4f ; sweeps %d vs %d" % (np.abs(J64 - J32).max() / np.abs(J64).max(), (res["float64"][1] != res["float32"][1]).mean(), res["float64"][2], res["float32"][2]))
