"""Pendulum swing-up by value iteration -- the script of pyro's
examples/demos_by_tool/lqr_vs_valueiteration_for_a_simple_pendulum.py / pendulum_optimal_swingup.py with the imports
switched to pyro_amd (BASELINE configs[0] at 101x101x11; pass a larger grid to taste, e.g. 1001 1001 51 float32).

    python examples/pendulum_optimal_swingup.py [nx nv nu [dtype]]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import pendulum
from pyro_amd.planning import discretizer, dynamicprogramming

nx, nv, nu = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (101, 101, 11)
dtype = sys.argv[4] if len(sys.argv) > 4 else "float64"

sys_ = pendulum.SinglePendulum()
grid_sys = discretizer.GridDynamicSystem(sys_, [nx, nv], [nu])

qcf = costfunction.QuadraticCostFunction.from_sys(sys_)
qcf.xbar = np.array([-3.14, 0])          # target: upright
qcf.INF = 300

dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid_sys, qcf, dtype=dtype)
dp.solve_bellman_equation(tol=0.1)       # 618 sweeps on the default grid, as the reference
dp.clean_infeasible_set()

ctl = dp.get_lookup_table_controller()
t, X, U = dp.simulate_closed_loop(np.array([[0.0, 0.0], [1.0, 0.0], [-2.0, 1.0]]), tf=10.0, n=2001)
for x0, xT in zip(X[:, 0], X[:, -1]):
    print("x0 = %s  ->  x(10 s) = %s   u(x0) = %+.3f" % (x0, np.round(xT, 3), ctl.c(x0, 0)[0]))

# ... and what the reference's script does next, unchanged: the closed loop of the policy from one initial state, with the plant's
# inputs and cost along it (one row of the same rollout kernel; pyro_amd/analysis/simulation.py)
cl_sys = ctl + sys_
cl_sys.x0 = np.array([0.0, 0.0])
traj = cl_sys.compute_trajectory(10, 2001, "euler")
print("closed loop from hanging at rest: x(10 s) = %s, cost along the trajectory J = %.2f" % (np.round(traj.x[-1], 3), traj.J[-1]))
