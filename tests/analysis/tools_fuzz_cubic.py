"""Randomised run of dp.interpol_method = 'cubic' (round 6: 2-D grids, the table tier's spline sweep with the in-box mask as validity
table) against the CPU twin's restatement (oracle/vi_oracle.py sweep_lut(method='cubic') / sweep_base_cubic: scipy's exact
'cubic_legacy' interpolant, zero outside the box): random pendulum-family problems, dims, action counts, bounds, dt, alpha, sweep
counts, LUT class and cell-by-cell base class, float64 and float32 storage.  Under emulation today, on a GPU once one is reachable.

usage: tools_fuzz_cubic.py [n_cases] [seed]"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))      # the repository root
import numpy as np

from oracle import vi_oracle as O
from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import pendulum
from pyro_amd.planning import discretizer
from pyro_amd.planning import dynamicprogramming as DP

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails, worst = 0, {"float64": 0.0, "float32": 0.0}
for case in range(n_cases):
    with contextlib.redirect_stdout(io.StringIO()):
        inverted = rng.random() < 0.4
        s = pendulum.InvertedPendulum() if inverted else pendulum.SinglePendulum()
        dims = [int(rng.integers(4, 140)), int(rng.integers(4, 140))]
        udims = [int(rng.integers(1, 24))]
        s.x_ub, s.x_lb = s.x_ub * rng.uniform(0.3, 1.5, size=2), s.x_lb * rng.uniform(0.3, 1.5, size=2)
        dt = float(rng.choice([0.01, 0.05, 0.1]))
        grid = discretizer.GridDynamicSystem(s, dims, udims, dt=dt)
        cf = costfunction.QuadraticCostFunction.from_sys(s)
        cf.xbar = rng.uniform(s.x_lb, s.x_ub) * 0.5
        cf.INF = float(rng.choice([50.0, 300.0, 1000.0]))
        alpha, nsw = float(rng.choice([1.0, 0.97])), int(rng.integers(1, 7))
        base = bool(rng.random() < 0.3)
        dtype = str(rng.choice(["float64", "float64", "float32"]))
        dp = (DP.DynamicProgramming if base else DP.DynamicProgrammingWithLookUpTable)(grid, cf, dtype=dtype)
        dp.save_time_history = False
        dp.alpha = alpha
        dp.interpol_method = str(rng.choice(["cubic", "cubic_legacy"]))
        dp.compute_steps(nsw)
        dyn_id, params = s.device_dynamics()
        p = O.Problem(grid.x_level, grid.u_level, dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar, float(cf.INF), float(cf.EPS),
                      x_lb=s.x_lb, x_ub=s.x_ub, u_lb=s.u_lb, u_ub=s.u_ub)
        xn, x_ok, a_ok, G = O.cells(p, np.arange(p.nodes_n))
        J = O.terminal_cost(p)
        for _ in range(nsw):
            if base:
                J, pi, Q = O.sweep_base_cubic(p.levels, xn, G, x_ok & a_ok, J, float(cf.INF), alpha)
            else:
                J, pi, Q = O.sweep_lut(p.levels, xn, G, J, alpha, method="cubic")
            if dtype == "float32":
                J = J.astype(np.float32).astype(np.float64)
    e = np.abs(dp.J - J).max() / max(np.abs(J).max(), 1e-300)
    tol = 1e-10 if dtype == "float64" else 2e-6
    Qs = np.sort(Q, axis=1)
    clear = (Qs[:, 1] - Qs[:, 0] > 1e-6 * max(np.abs(J).max(), 1.0)) if Q.shape[1] > 1 else np.ones(len(J), bool)
    bad = e > tol or not np.array_equal(dp.pi[clear], pi[clear]) or not dp._p.describe().startswith("path=spline")
    worst[dtype] = max(worst[dtype], e)
    fails += bool(bad)
    print("%3d %-8s dims %-10s A %-4s dt %.2f a %.2f sw %d %-4s %-7s err %.2e %s" % (case, "inverted" if inverted else "pendulum", dims, udims, dt, alpha, nsw,
                                                                                   "base" if base else "lut", dtype, e, "FAIL " + dp._p.describe()[:100] if bad else ""), flush=True)
print("cubic: failures %d / %d; worst errors %s" % (fails, n_cases, {k: "%.2e" % v for k, v in worst.items()}))
sys.exit(1 if fails else 0)
