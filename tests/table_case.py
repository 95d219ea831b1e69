"""A system the library has no kernel for (arbitrary Python f, a validity test that is not a box): the table tier.
Used by the sharded table-tier test on both sides (rank processes and the single-GPU reference)."""
import numpy as np

from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import system
from pyro_amd.planning import discretizer


class DuffingCart(system.ContinuousDynamicSystem):
    def __init__(self):
        super().__init__(2, 1, 2)
        self.name = "Duffing cart"
        self.x_ub, self.x_lb = np.array([3.0, 4.0]), np.array([-3.0, -4.0])
        self.u_ub, self.u_lb = np.array([6.0]), np.array([-6.0])

    def f(self, x, u, t=0):
        return np.array([x[1], u[0] - 0.5 * x[1] - x[0] ** 3 + x[0]])

    def isavalidstate(self, x):
        return bool(system.ContinuousDynamicSystem.isavalidstate(self, x) and (x[0] ** 2 + (x[1] - 2.5) ** 2 > 0.6))


def table_case():
    s = DuffingCart()
    g = discretizer.GridDynamicSystem(s, [37, 29], [5], 0.05)
    cf = costfunction.QuadraticCostFunction.from_sys(s)
    cf.xbar = np.array([1.0, 0.0])
    cf.INF = 200.0
    cf.S = np.diag([3.0, 1.0])
    return dict(grid_sys=g, cf=cf, dtype="float64")
