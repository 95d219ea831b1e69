"""Helicopter in a tunnel with obstacle boxes -- pyro's 3-D dynamic-programming demo
(examples/demos_by_tool/dynamicprogramming/helicopter_tunnel.py) with the imports switched to pyro_amd.  The obstacle test
(isavalidstate) and the domain-checked quadratic cost run in-kernel; float32 takes the mask-walking sweep k_sweep3_fast.

    python examples/helicopter_tunnel.py [n [dtype]]        # n grid levels per axis (the demo: 51), float64 | float32
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import drone
from pyro_amd.planning import discretizer, dynamicprogramming

n = int(sys.argv[1]) if len(sys.argv) > 1 else 51
dtype = sys.argv[2] if len(sys.argv) > 2 else "float64"

heli = drone.ConstantSpeedHelicopterTunnel()
heli.obstacles = [[(2, 2), (4, 4)], [(8, 5), (10, 10)], [(14, 0), (16, 4)]]
heli.mass, heli.vx, heli.width = 0.1, 5.0, 1.0
heli.x_ub, heli.x_lb = np.array([+60.0, 10.0, +20.0]), np.array([-60.0, 0.0, 0.0])
heli.u_ub, heli.u_lb = np.array([+20.0]), np.array([-20.0])

grid_sys = discretizer.GridDynamicSystem(heli, (n, n, n), [11], 0.05)

qcf = costfunction.QuadraticCostFunctionWithDomainCheck.from_sys(heli)
qcf.xbar = np.array([0.0, 2.0, 20.0])        # fly low, reach the end of the tunnel
qcf.INF, qcf.EPS = 100000, 0.2
qcf.Q[0, 0], qcf.Q[1, 1], qcf.Q[2, 2] = 2.0, 200.0, 0.0
qcf.R[0, 0] = 5.0
qcf.S[0, 0], qcf.S[1, 1], qcf.S[2, 2] = 20.0, 50.0, 0.0

dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid_sys, qcf, dtype=dtype)
dp.alpha = 0.999
dp.solve_bellman_equation()                  # tol = 0.1
print("kernel:", dp._p.describe().split(" note=")[0])

ctl = dp.get_lookup_table_controller()
t, X, U = dp.simulate_closed_loop(np.array([[0.0, 6.0, 0.0], [0.0, 1.0, 0.0]]), tf=4.0, n=801)    # Euler roll-outs on the GPU
for x0, xT in zip(X[:, 0], X[:, -1]):
    print("x0 = %s  ->  x(4 s) = %s   u(x0) = %+.2f" % (x0, np.round(xT, 2), ctl.c(x0, 0)[0]))
