"""CPU restatement of error-feedback storage (PVI_FLAG_F32_FEEDBACK, sweep_lean4.inc lean4_feedback) on the oracle's C twin:
float64 arithmetic with (a) plain float32 storage of J, (b) float32 storage + a float32 residual per node fed back into the
node's next store, (c) the same with a float16 residual.  Cart-pole 31^4 x 21 by default (argv[1]: another workload of
pyro_amd.configs, argv[2]: sweeps, e.g. `pendulum:201,201:21:float32 2000` for the 2-D form lean_feedback); prints
max|J - J64| / max|J64| every 50 sweeps.
Test infrastructure (imports oracle/): not on any product path."""
import contextlib, io, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from pyro_amd import configs
from oracle import vi_oracle as O, c_oracle as CO
with contextlib.redirect_stdout(io.StringIO()):
    cfg = configs.build(sys.argv[1] if len(sys.argv) > 1 else "cartpole:31,31,31,31:21:float32")
s, g, cf = cfg["sys"], cfg["grid_sys"], cfg["cf"]
dyn_id, params = s.device_dynamics()
p = O.Problem(g.x_level, g.u_level, g.dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar, float(cf.INF), float(cf.EPS))
c = CO.CProblem(p)
J64 = c.terminal_cost().copy()
Ja = J64.copy()            # plain f32 storage
Jb = J64.copy(); lob = np.zeros_like(Jb)   # feedback, f32 residual
Jc = J64.copy(); loc = np.zeros_like(Jc)   # feedback, fp16 residual
t0 = time.time()
def one(J):
    out, _, _ = c.sweeps(J, 1, threads=8)
    return out.copy()
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 800
for k in range(1, NS + 1):
    J64 = one(J64)
    Ja = one(Ja).astype(np.float32).astype(np.float64)
    t = one(Jb) + lob; Jb = t.astype(np.float32).astype(np.float64); lob = (t - Jb).astype(np.float32).astype(np.float64)
    t = one(Jc) + loc; Jc = t.astype(np.float32).astype(np.float64); loc = (t - Jc).astype(np.float16).astype(np.float64)
    if k % max(50, NS // 16) == 0:
        m = np.abs(J64).max()
        print(k, "plain %.3e  fb32 %.3e  fb16 %.3e" % (np.abs(Ja - J64).max() / m, np.abs(Jb - J64).max() / m, np.abs(Jc - J64).max() / m), "%.0fs" % (time.time() - t0), flush=True)
