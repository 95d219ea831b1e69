"""Host mirror of pyro/dynamic/suspension.py:20-124 (QuarterCarOnRoughTerrain): a sprung mass driven at constant
speed over a sum-of-sines ground profile, x = [dy, y, x], u = [f].  Drawing code is out of scope."""
import numpy as np

from pyro_amd import _native
from pyro_amd.dynamic import system


class QuarterCarOnRoughTerrain(system.ContinuousDynamicSystem):

    def __init__(self):
        super().__init__(3, 1, 3)
        self.name = "Quarter Car on Rought Terrain"
        self.state_label, self.state_units = ["dy", "y", "x"], ["[m/sec]", "[m]", "[m]"]
        self.input_label, self.input_units = ["f"], ["[N]"]
        self.output_label, self.output_units = self.state_label, self.state_units
        self.x_ub = np.array([+10, +10, +10])
        self.x_lb = np.array([-10, -10, -10])
        self.mass, self.b, self.k, self.vx = 1, 1, 1, 1
        self.dynamic_domain, self.dynamic_range = True, 10
        self.a = np.array([0.5, 0.3, 0.7, 0.2, 0.2, 0.1])       # amplitude
        self.w = np.array([0.2, 0.4, 0.5, 1.0, 2.0, 3.0])       # spatial frequency
        self.phi = np.array([3.0, 2.0, 0.0, 0.0, 0.0, 0.0])     # phase

    def z(self, x):
        """Ground level at x (suspension.py:72-80)."""
        z = 0
        for i in range(self.a.size):
            z = z + self.a[i] * np.sin(self.w[i] * (x - self.phi[i]))
        return z

    def dz(self, x):
        """Ground slope at x (suspension.py:84-92)."""
        dz = 0
        for i in range(self.a.size):
            dz = dz + self.a[i] * self.w[i] * np.cos(self.w[i] * (x - self.phi[i]))
        return dz

    def f(self, x=np.zeros(3), u=np.zeros(1), t=0):
        dx = np.zeros(self.n)
        z = self.z(x[2])
        dz = self.dz(x[2])
        dx[0] = 1. / self.mass * (u[0] - self.k * (x[1] - z) - self.b * (x[0] - dz))
        dx[1] = x[0]
        dx[2] = self.vx
        return dx

    def xut2q(self, x, u, t):
        return np.array([x[2], x[1]])

    # ---- device path: closed form PVI_DYN_QUARTERCAR; the ground profile comes from the system's OWN z / dz, evaluated
    # at the levels of axis 2 (a subclass with another terrain keeps the kernel) -------------------------------------
    def device_dynamics(self):
        if not self.stock_model(QuarterCarOnRoughTerrain, ("f",)):
            return None
        return _native.DYN_QUARTERCAR, [1. / self.mass, float(self.k), float(self.b), float(self.vx)]

    def device_rollout_params(self):
        """Constants of the continuous closed form for GPU rollouts (pvi_set_rollout_params): the stock sum-of-sines ground
        [terms, a, w, phi]; a subclass with its own z / dz gets None -> the host loop."""
        if not self.stock_model(QuarterCarOnRoughTerrain, ("f", "z", "dz")) or 1 + 3 * self.a.size > 64:
            return None
        return np.concatenate([[float(self.a.size)], self.a, self.w, self.phi]).astype(float)

    def device_trig(self, x_level):
        lv = x_level[2]
        return (np.array([self.z(lv[i]) for i in range(len(lv))], dtype=float),
                np.array([self.dz(lv[i]) for i in range(len(lv))], dtype=float))
