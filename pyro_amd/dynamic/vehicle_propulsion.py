"""Host mirror of pyro/dynamic/vehicle_propulsion.py:23-223 (LongitudinalFrontWheelDriveCarWithWheelSlipInput): a car
on a straight road driven by the slip ratio of its front wheels, x = [x, dx], u = [slip]; weight transfer couples the
traction to the acceleration, and inputs that would lift a wheel are not allowed.  Drawing code is out of scope."""
import numpy as np

from pyro_amd import _native
from pyro_amd.dynamic import system


class LongitudinalFrontWheelDriveCarWithWheelSlipInput(system.ContinuousDynamicSystem):

    def __init__(self):
        super().__init__(2, 1, 2)
        self.name = "Front Wheel Drive Car"
        self.state_label, self.state_units = ["x", "dx"], ["[m]", "[m/sec]"]
        self.input_label, self.input_units = ["slip"], ["[]"]
        self.output_label, self.output_units = ["x", "dx"], ["[m]", "[m/sec]"]
        self.x_ub = np.array([+50, +30, ])
        self.x_lb = np.array([0, -10])
        self.u_ub = np.array([0.3])
        self.u_lb = np.array([-0.3])
        self.lenght, self.xc, self.yc = 2, 1, 0.5          # wheel base, c.g. behind the front... (the reference's names)
        self.mass, self.gravity = 1500, 9.81
        self.rho, self.cdA = 1.225, 0.3 * 2
        self.mu_max, self.mu_slope = 1.0, 70.
        self.dynamic_domain, self.dynamic_range = False, self.lenght * 2
        self.linestyle = "-"
        self.obs_dist = self.x_ub[0] + self.lenght * 2
        self.obs_size = 2

    def compute_ratios(self):
        ry = self.yc / self.lenght
        rr = self.xc / self.lenght
        rf = 1 - rr
        return ry, rr, rf

    def slip2force(self, slip):
        """Ground traction curve (sigmoid), vehicle_propulsion.py:96-102."""
        return self.mu_max * (2 / (1 + np.exp(-self.mu_slope * slip)) - 1)

    def _acceleration(self, x, u):
        slip = u
        v = x[1]
        mu = self.slip2force(slip)
        ry, rr, rf = self.compute_ratios()
        m, g = self.mass, self.gravity
        rcda = self.rho * self.cdA
        fd = 0.5 * rcda * v * np.abs(v)
        a = (mu * m * g * rr - fd) / (m * (1 + mu * ry))
        return a, (m, g, ry, rr, rf)

    def f(self, x, u, t=0):
        dx = np.zeros(self.n)
        a, _ = self._acceleration(x, u)
        dx[0] = x[1]
        dx[1] = a
        return dx

    def isavalidinput(self, x, u):
        """Slip bounds, and no negative normal force on either axle (vehicle_propulsion.py:186-223)."""
        ans = False
        for i in range(self.m):
            ans = ans or (u[i] < self.u_lb[i])
            ans = ans or (u[i] > self.u_ub[i])
        a, (m, g, ry, rr, rf) = self._acceleration(x, u)
        fn_front = m * g * rr - m * a * ry
        fn_rear = m * g * rf + m * a * ry
        ans = ans or (fn_front < 0.)
        ans = ans or (fn_rear < 0.)
        return not bool(np.all(ans))

    def xut2q(self, x, u, t):
        return np.append(x, u[0])

    # ---- device path: closed form PVI_DYN_LONGCAR; the wheel-load test of isavalidinput runs per cell in-kernel ------
    _INPUT_VALIDITY_OWNER = None

    def device_dynamics(self):
        if not self.stock_model(LongitudinalFrontWheelDriveCarWithWheelSlipInput,
                                ("f", "_acceleration", "slip2force", "compute_ratios")):
            return None
        ry, rr, rf = self.compute_ratios()
        m, g = self.mass, self.gravity
        return _native.DYN_LONGCAR, [float(m), float(ry), float(m * g * rr), float(m * g * rf)]

    def device_rollout_params(self):
        """Constants of the continuous closed form for GPU rollouts: [mu_max, mu_slope, rho cdA, m, g, ry, rr]."""
        ry, rr, rf = self.compute_ratios()
        return np.array([self.mu_max, self.mu_slope, self.rho * self.cdA, self.mass, self.gravity, ry, rr], dtype=float)

    def device_trig(self, x_level):
        rcda = self.rho * self.cdA
        v = x_level[1]
        return (np.array([0.5 * rcda * v[i] * np.abs(v[i]) for i in range(len(v))], dtype=float),)

    def device_act_aux(self, input_from_action_id):
        """Per action {mu m g rr, m (1 + mu ry)}: the two action-dependent factors of the acceleration, evaluated
        here in NumPy with the reference's own expressions."""
        ry, rr, rf = self.compute_ratios()
        m, g = self.mass, self.gravity
        U = np.asarray(input_from_action_id, dtype=float)
        out = np.empty((U.shape[0], 2))
        for a in range(U.shape[0]):
            mu = self.slip2force(U[a, 0])
            out[a] = mu * m * g * rr, m * (1 + mu * ry)
        return out


LongitudinalFrontWheelDriveCarWithWheelSlipInput._INPUT_VALIDITY_OWNER = LongitudinalFrontWheelDriveCarWithWheelSlipInput
