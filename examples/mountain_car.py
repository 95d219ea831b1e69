"""pyro's examples/demos_by_system/mountain_car/mountain_car_with_valueiteration_quadratic.py with pyro_amd imports.
MountainCar has position-dependent inertia / Coriolis / actuator / gravity terms and no closed-form kernel: it runs
through the generic mechanical tier (per-node tables, O(N) evaluations of the model terms)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import mountaincar
from pyro_amd.planning import discretizer, dynamicprogramming

sys_ = mountaincar.MountainCar()
sys_.x_ub, sys_.x_lb = np.array([+0.2, +2.0]), np.array([-1.7, -2.0])
sys_.u_ub[0], sys_.u_lb[0] = +0.2, -0.2

grid_sys = discretizer.GridDynamicSystem(sys_, [201, 201], [11])

qcf = costfunction.QuadraticCostFunction.from_sys(sys_)
qcf.xbar = np.array([0, 0])
qcf.INF = 30
qcf.R[0, 0] = 10.0
qcf.S[0, 0] = 10.0
qcf.S[1, 1] = 10.0

dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid_sys, qcf, dtype="float32")
print("tier:", dp.tier, "|", dp._p.describe().split()[0])
dp.solve_bellman_equation(tol=0.01)
dp.clean_infeasible_set()
ctl = dp.get_lookup_table_controller()
print("u(-1, 0) =", ctl.c(np.array([-1.0, 0.0]), 0))
