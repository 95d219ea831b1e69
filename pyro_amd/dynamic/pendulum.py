"""
Host mirror of the pendulum systems used by the DP path (reference pyro/dynamic/pendulum.py:
SinglePendulum :16, InvertedPendulum :283, DoublePendulum :340).  Drawing code is out of scope.
"""
import numpy as np

from pyro_amd import _native
from pyro_amd.dynamic import mechanical


class SinglePendulum(mechanical.MechanicalSystem):
    """Point mass m1 at lc1 on a rod of inertia I1; q = 0 is hanging down (pendulum.py:52-66)."""

    _gravity_sign = 1.0

    def __init__(self):
        super().__init__(1)
        self.name = "Single Pendulum"
        self.setparams()

    def setparams(self):
        self.l1, self.lc1 = 2.0, 1
        self.m1, self.I1, self.gravity, self.d1 = 1, 1, 9.81, 0
        self.l_domain = 5.0

    def H(self, q):
        return np.array([[self.m1 * self.lc1 ** 2 + self.I1]], dtype=float)

    def C(self, q, dq):
        return np.zeros((1, 1))

    def B(self, q):
        return np.eye(1)

    def g(self, q):
        return np.array([self._gravity_sign * self.m1 * self.gravity * self.lc1 * np.sin(q[0])])

    def d(self, q, dq):
        return np.array([self.d1 * dq[0]])

    # device: c = [1/H, signed m1*g*lc1, d1]   (kernel: Dyn<PVI_DYN_PENDULUM>).  The kernel hard-codes THIS class's
    # H, C, B, g, d: a subclass or instance that overrides any of them runs through the per-node tables instead.
    def _closed_form(self):
        return self.stock_model(SinglePendulum, self._MODEL_TERMS)

    def device_dynamics(self):
        if not self._closed_form():
            return mechanical.MechanicalSystem.device_dynamics(self)
        H = self.m1 * self.lc1 ** 2 + self.I1
        gc = self.m1 * self.gravity * self.lc1
        return _native.DYN_PENDULUM, [1.0 / float(H), self._gravity_sign * gc, float(self.d1), float(H)]

    def device_trig(self, x_level):
        if not self._closed_form():
            return mechanical.MechanicalSystem.device_trig(self, x_level)
        return (np.sin(x_level[0]),)


class InvertedPendulum(SinglePendulum):
    """Same model with gravity flipped: q = 0 is upright (pendulum.py:283-312)."""

    _gravity_sign = -1.0

    def __init__(self):
        super().__init__()
        self.name = "Inverted Pendulum"


class _TwoLinkTerms:
    """H, C, g, d shared by DoublePendulum (pendulum.py:400-493) and TwoLinkManipulator
    (manipulator.py:897-992); B = I."""

    def _trig(self, q):
        return np.cos(q[1]), np.sin(q[1]), np.sin(q[0]), np.sin(q[0] + q[1])

    def H(self, q):
        c2 = np.cos(q[1])
        H01 = self.m2 * self.lc2 ** 2 + self.m2 * self.l1 * self.lc2 * c2 + self.I2
        H00 = (self.m1 * self.lc1 ** 2 + self.I1
               + self.m2 * (self.l1 ** 2 + self.lc2 ** 2 + 2 * self.l1 * self.lc2 * c2) + self.I2)
        return np.array([[H00, H01], [H01, self.m2 * self.lc2 ** 2 + self.I2]], dtype=float)

    def C(self, q, dq):
        h = self.m2 * self.l1 * self.lc2 * np.sin(q[1])
        return np.array([[-h * dq[1], -h * (dq[0] + dq[1])], [h * dq[0], 0.0]])

    def B(self, q):
        return np.eye(2)

    def g(self, q):
        g1 = (self.m1 * self.lc1 + self.m2 * self.l1) * self.gravity
        g2 = self.m2 * self.lc2 * self.gravity
        s1, s12 = np.sin(q[0]), np.sin(q[0] + q[1])
        return np.array([-g1 * s1 - g2 * s12, -g2 * s12])

    def d(self, q, dq):
        return np.array([self.d1 * dq[0], self.d2 * dq[1]])

    # device: c = [k0, m2, k1, k2, I2, k3, k4, g1c, g2c, d1, d2]   (kernel: Dyn<PVI_DYN_TWOLINK>).  The kernel
    # hard-codes the terms above with B = I and no end-effector force: any override (H, C, B, g, d, ddq, f, and for
    # manipulators J / f_ext) sends the system to the per-node tables of the generic mechanical tier.
    _STOCK_OWNER = None         # set below: the concrete class whose model the kernel implements

    def _closed_form(self):
        names = self._MODEL_TERMS + (("J", "f_ext") if hasattr(self, "f_ext") else ())
        owner = next(c for c in type(self).__mro__ if c.__dict__.get("_STOCK_OWNER") is c)
        return self.stock_model(owner, names)

    def device_dynamics(self):
        if not self._closed_form():
            return mechanical.MechanicalSystem.device_dynamics(self)
        k0 = self.m1 * self.lc1 ** 2 + self.I1
        k1 = self.l1 ** 2 + self.lc2 ** 2
        k2 = 2 * self.l1 * self.lc2
        k3 = self.m2 * self.lc2 ** 2
        k4 = self.m2 * self.l1 * self.lc2
        g1c = (self.m1 * self.lc1 + self.m2 * self.l1) * self.gravity
        g2c = self.m2 * self.lc2 * self.gravity
        return _native.DYN_TWOLINK, [float(v) for v in
                                     (k0, self.m2, k1, k2, self.I2, k3, k4, g1c, g2c, self.d1, self.d2)]

    def device_trig(self, x_level):
        if not self._closed_form():
            return mechanical.MechanicalSystem.device_trig(self, x_level)
        q0, q1 = x_level[0], x_level[1]
        return np.sin(q0), np.cos(q1), np.sin(q1), np.sin(q0[:, None] + q1[None, :])


    # ---- per-node tables of the generic mechanical tier (a subclass with another actuator matrix: Acrobot) by array
    # arithmetic over the grid axes instead of one Python call per node (round 5)
    def _trig_vectorized(self, x_level):
        """a0 = inv(H)(-C dq - g - d), Bn = inv(H) B over the 4-D grid with H, C, g, d of this class on whole axes; inv(H)
        by adjugate / determinant like the closed-form kernel (the per-node loop calls LAPACK: an ulp apart, the known
        limit of DESIGN.md section 2).  None when a model term other than B is overridden."""
        owner = next((c for c in type(self).__mro__ if c.__dict__.get("_STOCK_OWNER") is c), None)
        if owner is None or not self.stock_model(owner, tuple(t for t in self._MODEL_TERMS if t != "B")):
            return None
        q0, q1, v0, v1 = (np.asarray(l, dtype=float) for l in x_level)
        c2, s2 = np.cos(q1), np.sin(q1)
        H01 = self.m2 * self.lc2 ** 2 + self.m2 * self.l1 * self.lc2 * c2 + self.I2
        H00 = (self.m1 * self.lc1 ** 2 + self.I1
               + self.m2 * (self.l1 ** 2 + self.lc2 ** 2 + 2 * self.l1 * self.lc2 * c2) + self.I2)
        H11 = self.m2 * self.lc2 ** 2 + self.I2
        det = H00 * H11 - H01 * H01
        i00, i01, i11 = H11 / det, -H01 / det, H00 / det                   # [n1]
        h = self.m2 * self.l1 * self.lc2 * s2                              # [n1]
        g1 = (self.m1 * self.lc1 + self.m2 * self.l1) * self.gravity
        g2 = self.m2 * self.lc2 * self.gravity
        s1 = np.sin(q0)[:, None]
        s12 = np.sin(q0[:, None] + q1[None, :])
        G0, G1 = -g1 * s1 - g2 * s12, -g2 * s12                            # [n0, n1]
        V0, V1 = v0[:, None], v1[None, :]                                  # [n2, 1], [1, n3]
        hh = h[None, :, None, None]
        # C dq = [-h dq1 dq0 - h (dq0 + dq1) dq1, h dq0 dq0]
        Cdq0 = (-hh * V1) * V0 + (-hh * (V0 + V1)) * V1
        Cdq1 = (hh * V0) * V0 + 0.0 * V1
        r0 = ((0.0 - Cdq0) - G0[:, :, None, None]) - self.d1 * V0
        r1 = ((0.0 - Cdq1) - G1[:, :, None, None]) - self.d2 * V1
        I00, I01, I11 = (a[None, :, None, None] for a in (i00, i01, i11))
        a0 = np.stack([I00 * r0 + I01 * r1, I01 * r0 + I11 * r1], axis=-1)
        B = np.asarray(self.B(np.zeros(2)), dtype=float)                   # (constant actuator matrices only)
        if not all(np.array_equal(np.asarray(self.B(np.array([a, b]))), B) for a, b in ((0.3, -1.1), (2.0, 0.7))):
            return None
        inv = np.stack([np.stack([i00, i01], -1), np.stack([i01, i11], -1)], -2)   # [n1, 2, 2]
        Bn = np.broadcast_to((inv @ B)[None], (len(q0),) + (len(q1), 2, B.shape[1])).reshape(-1, 2, B.shape[1])
        return a0.reshape(-1, 2), np.ascontiguousarray(Bn)


class DoublePendulum(_TwoLinkTerms, mechanical.MechanicalSystem):
    """Two unit links, torques at both joints (pendulum.py:340-378)."""

    def __init__(self):
        mechanical.MechanicalSystem.__init__(self, 2)
        self.name = "Double Pendulum"
        self.setparams()
        self.l_domain = 3

    def setparams(self):
        self.l1 = self.l2 = self.lc1 = self.lc2 = 1
        self.m1 = self.m2 = 1
        self.I1 = self.I2 = 0
        self.gravity = 9.81
        self.d1 = self.d2 = 0


DoublePendulum._STOCK_OWNER = DoublePendulum


class Acrobot(DoublePendulum):
    """Double pendulum with a single motor at the elbow (pendulum.py:699-735): B = [[0], [1]], |tau| <= 10.
    Under-actuated: value iteration runs through the per-node tables of the generic mechanical tier (PVI_DYN_NODE_2x1),
    filled by array arithmetic over the grid axes (_TwoLinkTerms._trig_vectorized) -- no Python call per node."""

    def __init__(self):
        mechanical.MechanicalSystem.__init__(self, dof=2, actuators=1)
        self.name = "Acrobot"
        self.input_label[0], self.input_units[0] = "tau", "[Nm]"
        self.u_lb[0], self.u_ub[0] = -10, +10
        self.setparams()
        self.l_domain = 3

    def B(self, q):
        return np.array([[0], [1]])

    # the closed-form two-link kernel assumes both joints are driven: use the generic mechanical tier
    def device_dynamics(self):
        return mechanical.MechanicalSystem.device_dynamics(self)

    def device_trig(self, x_level):
        return mechanical.MechanicalSystem.device_trig(self, x_level)
