"""Host mirror of pyro/dynamic/mountaincar.py:22-213 (MountainCar): a point mass sliding on the terrain
z(x) = a cos(w x), driven along the slope.  A one-degree-of-freedom Manipulator whose inertia, Coriolis, actuator and
gravity terms depend on the position: value iteration runs through the per-node tables of the generic mechanical tier
(PVI_DYN_NODE_1x1), which the STOCK model fills by array arithmetic over the grid axes (`_trig_vectorized`: the operations
of MechanicalSystem.ddq, mechanical.py:238-263, in its order -- no Python call per node); a subclass that overrides a model
term falls back to the per-node loop of MechanicalSystem.device_trig.  Drawing code is out of scope."""
import numpy as np

from pyro_amd.dynamic import manipulator


class MountainCar(manipulator.Manipulator):

    def __init__(self):
        manipulator.Manipulator.__init__(self, 1, 1, 2)
        self.name = "Mountain Car"
        self.state_label, self.state_units = ["x", "dx"], ["[m]", "[m/sec]"]
        self.input_label, self.input_units = ["throttle"], ["[N]"]
        self.output_label, self.output_units = ["x", "dx"], ["[m]", "[m/sec]"]
        # mountaincar.py:57-61
        self.x_ub, self.x_lb = np.array([0.2, 0.5]), np.array([-1.7, -0.5])
        self.u_ub, self.u_lb = np.array([1.0]), np.array([-1.0])
        # mountaincar.py:64-70
        self.mass, self.gravity = 1.0, 1.0
        self.a, self.w = 0.5, np.pi

    # terrain profile and its derivatives (mountaincar.py:82-110)
    def z(self, x):
        return self.a * np.cos(self.w * x)

    def dz_dx(self, x):
        return -self.a * self.w * np.sin(self.w * x)

    def d2z_dx2(self, x):
        return -self.a * self.w ** 2 * np.cos(self.w * x)

    def forward_kinematic_effector(self, q):
        return np.array([q[0], self.z(q[0])])

    def J(self, q):
        J = np.zeros((self.e, self.dof))
        J[0] = 1
        J[1] = self.dz_dx(q[0])
        return J

    # model terms (mountaincar.py:129-213)
    def H(self, q):
        return np.array([[self.mass * (1 + self.dz_dx(q[0]) ** 2)]])

    def C(self, q, dq):
        return np.array([[self.mass * self.dz_dx(q[0]) * self.d2z_dx2(q[0]) * dq[0]]])

    def B(self, q):
        return np.array([[np.sqrt(1 + self.dz_dx(q[0]) ** 2)]])

    def g(self, q):
        return np.array([self.mass * self.gravity * self.dz_dx(q[0])])

    def d(self, q, dq):
        return np.zeros(1)

    # ---- the per-node tables of the generic mechanical tier without a Python call per node (round 5) ------------------
    def _trig_vectorized(self, x_level):
        """a0 = ddq(q, dq, 0), Bn = inv(H) B over the grid, by the expressions of the methods above on whole axes, in the
        order MechanicalSystem.ddq evaluates them (rhs = B u - C dq - g - d with u = 0; inv of a 1 x 1 matrix is 1 / h):
        the same bits as the per-node loop (tests/test_abi_cpu.py)."""
        if not self.stock_model(MountainCar, self._MODEL_TERMS + ("z", "dz_dx", "d2z_dx2")):
            return None
        x, v = np.asarray(x_level[0], dtype=float), np.asarray(x_level[1], dtype=float)
        dz, ddz = self.dz_dx(x), self.d2z_dx2(x)
        h = self.mass * (1 + dz ** 2)
        b = np.sqrt(1 + dz ** 2)
        g = self.mass * self.gravity * dz
        hinv = 1.0 / h
        c = (self.mass * dz * ddz)[:, None] * v[None, :]                    # C[0, 0](q, dq)
        rhs = ((b * 0.0)[:, None] - c * v[None, :]) - g[:, None] - 0.0     # B u - C dq - g - d
        a0 = hinv[:, None] * rhs
        Bn = (hinv * b).reshape(-1, 1, 1)
        return a0.reshape(-1, 1), Bn
