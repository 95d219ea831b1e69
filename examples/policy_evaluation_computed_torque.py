"""Cost-to-go of a GIVEN controller -- pyro's examples/demos_by_tool/dynamicprogramming/policy_evaluator_with_computed_torque.py
with the imports switched to pyro_amd: a computed-torque law on the pendulum, evaluated on a 301 x 301 grid.  The x_next / G
tables of the policy are built in ONE kernel (the controller is evaluated in-kernel, its inputs bit for bit the reference's);
the sweeps are the table-tier kernel with one action per node.

    python examples/policy_evaluation_computed_torque.py [n]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

from pyro_amd.analysis import costfunction
from pyro_amd.control import nonlinear
from pyro_amd.dynamic import pendulum
from pyro_amd.planning import discretizer, dynamicprogramming

n = int(sys.argv[1]) if len(sys.argv) > 1 else 301

pend = pendulum.SinglePendulum()
pend.x_ub, pend.x_lb = np.array([+6.0, +6.0]), np.array([-9.0, -6.0])
pend.u_ub[0], pend.u_lb[0] = +200.0, -200.0

ctl = nonlinear.ComputedTorqueController(pend)
ctl.rbar = np.array([-3.14])                 # target

qcf = costfunction.QuadraticCostFunction.from_sys(pend)
qcf.xbar = np.array([ctl.rbar[0], 0.0])
qcf.INF = 300
qcf.S[0, 0] = qcf.S[1, 1] = 10.0

grid_sys = discretizer.GridDynamicSystem(pend, [n, n], [11], 0.05, False)   # (no look-up tables of the open-loop system)

evaluator = dynamicprogramming.PolicyEvaluatorWithLookUpTable(ctl, grid_sys, qcf)
print("policy tables built on:", evaluator.tables_on)
evaluator.solve_bellman_equation()
J = evaluator.J.reshape(n, n)
print("J(hanging, at rest) = %.3f   J(target) = %.3f   max J = %.1f" % (J[np.abs(grid_sys.x_level[0]).argmin(), n // 2],
                                                                     J[np.abs(grid_sys.x_level[0] + 3.14).argmin(), n // 2], J.max()))
