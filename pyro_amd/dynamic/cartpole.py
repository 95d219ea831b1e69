"""Host mirror of CartPole (reference pyro/dynamic/cartpole.py:322-437): cart m1 on a rail with an
un-actuated pole (point mass m2 at lcg); theta = 0 is hanging down, upright is theta = pi."""
import numpy as np

from pyro_amd import _native
from pyro_amd.dynamic import mechanical


class CartPole(mechanical.MechanicalSystem):

    def __init__(self):
        super().__init__(dof=2, actuators=1)
        self.name = "Cart Pole"
        self.state_label[0], self.state_label[2] = "x", "dx"
        self.state_units[0], self.state_units[2] = "[m]", "[m/s]"
        self.input_label[0], self.input_units[0] = "F", "[N]"
        self.u_lb[0], self.u_ub[0] = -10, +10            # cartpole.py:349-350
        self.l, self.lcg = 3, 0.5
        self.m1, self.m2, self.gravity = 1, 0.1, 9.81

    def H(self, q):
        off = self.m2 * self.lcg * np.cos(q[1])
        return np.array([[self.m1 + self.m2, off], [off, self.m2 * self.lcg ** 2]], dtype=float)

    def C(self, q, dq):
        C = np.zeros((2, 2))
        C[0, 1] = -self.m2 * self.lcg * np.sin(q[1]) * dq[1]
        return C

    def B(self, q):
        return np.array([[1.0], [0.0]])

    def g(self, q):
        return np.array([0.0, self.m2 * self.gravity * self.lcg * np.sin(q[1])])

    def d(self, q, dq):
        return np.zeros(2)

    # device: c = [m1+m2, m2*lcg, m2*lcg^2, -m2*lcg, m2*g*lcg]   (kernel: Dyn<PVI_DYN_CARTPOLE>)
    def _closed_form(self):
        return self.stock_model(CartPole, self._MODEL_TERMS)

    def device_dynamics(self):
        if not self._closed_form():          # overridden model terms: per-node tables of the generic tier
            return mechanical.MechanicalSystem.device_dynamics(self)
        return _native.DYN_CARTPOLE, [float(self.m1 + self.m2), float(self.m2 * self.lcg),
                                      float(self.m2 * self.lcg ** 2), float(-self.m2 * self.lcg),
                                      float(self.m2 * self.gravity * self.lcg)]

    def device_trig(self, x_level):
        if not self._closed_form():
            return mechanical.MechanicalSystem.device_trig(self, x_level)
        return np.cos(x_level[1]), np.sin(x_level[1])
