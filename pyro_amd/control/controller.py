"""
The slice of pyro/control/controller.py the DP path hands its result to: StaticController
(controller.py:22) and the `controller + system` composition (ClosedLoopSystem, :248-355).
Everything else in that module (dynamic controllers, plots, simulation shortcuts) is out of scope.
"""
import numpy as np

from pyro_amd.dynamic import system


class StaticController:
    """u = c(y, r, t) with r in R^k, u in R^m, y in R^p."""

    def __init__(self, k=1, m=1, p=1):
        self.k, self.m, self.p = k, m, p
        self.name = "Static Controller"
        self.rbar = np.zeros(k)
        self.ref_label = ["Ref. %d" % i for i in range(k)]
        self.ref_units = [""] * k
        self.r_ub, self.r_lb = np.full(k, 10.0), np.full(k, -10.0)

    def c(self, y, r, t=0):
        raise NotImplementedError

    def t2r(self, t):
        return self.rbar

    def cbar(self, y, t=0):
        return self.c(y, self.t2r(t), t)

    def __add__(self, sys):
        return ClosedLoopSystem(sys, self)


class ClosedLoopSystem(system.ContinuousDynamicSystem):
    """dx = plant.f(x, controller.c(plant.h(x, ubar, t), r, t), t)  (controller.py:328-355)."""

    def __init__(self, plant, controller):
        if plant.p != controller.p:
            raise NameError("Dimension mismatch between controller and plant outputs")
        if plant.m != controller.m:
            raise NameError("Dimension mismatch between controller and plant inputs")
        self.plant, self.controller = plant, controller
        super().__init__(plant.n, controller.k, plant.p)
        self.name = "Closed-Loop " + plant.name + " with " + controller.name
        self.state_label, self.state_units = plant.state_label, plant.state_units
        self.x_ub, self.x_lb = plant.x_ub, plant.x_lb
        self.u_ub, self.u_lb = controller.r_ub, controller.r_lb
        self.xbar, self.ubar = plant.xbar, controller.rbar
        self.x0 = plant.x0

    def f(self, x, u, t=0):
        y = self.plant.h(x, self.plant.ubar, t)
        return self.plant.f(x, self.controller.c(y, u, t), t)

    def h(self, x, u, t=0):
        return self.plant.h(x, self.plant.ubar, t)
