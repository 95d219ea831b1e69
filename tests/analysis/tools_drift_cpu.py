import contextlib, io, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from pyro_amd import configs
from oracle import vi_oracle as O, c_oracle as CO
with contextlib.redirect_stdout(io.StringIO()):
    cfg = configs.build("cartpole:31,31,31,31:21:float32")
s, g, cf = cfg["sys"], cfg["grid_sys"], cfg["cf"]
dyn_id, params = s.device_dynamics()
p = O.Problem(g.x_level, g.u_level, g.dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar, float(cf.INF), float(cf.EPS))
c = CO.CProblem(p)
J64 = c.terminal_cost(); J32 = J64.copy()
t0 = time.time(); k = 0
while k < 1200:
    J64, _, _ = c.sweeps(J64, 50, threads=8)
    J64 = J64.copy()
    J32, _, _ = c.sweeps(J32, 50, f32=True, threads=8)
    J32 = J32.copy()
    k += 50
    print(k, "%.3e" % (np.abs(J32 - J64).max() / np.abs(J64).max()), "maxJ %.2f" % J64.max(), "mean signed %.3e" % ((J32 - J64).mean() / np.abs(J64).max()), "%.0fs" % (time.time() - t0), flush=True)
