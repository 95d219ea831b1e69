"""Host mirror of the kinematic bicycle models of pyro/dynamic/vehicle_steering.py: KinematicBicyleModel (:20-86),
KinematicCarModel (:717-749, a real-sized car) and KinematicCarModelwithObstacles (:973-1021, the car-parking demo).
x = [x, y, theta], u = [v, beta] (speed, steering angle).  Drawing code is out of scope."""
import numpy as np

from pyro_amd import _native
from pyro_amd.dynamic import system


class HolonomicMobileRobot(system.ContinuousDynamicSystem):
    """Holonomic 2-D point robot: dx = u0, dy = u1 (vehicle_steering.py:201-259)."""

    def __init__(self):
        super().__init__(2, 2, 2)
        self.name = "Holonomic Mobile Robot"
        self.state_label, self.state_units = ["x", "y"], ["[m]", "[m]"]
        self.input_label, self.input_units = ["vx", "vy"], ["[m/sec]", "[m/sec]"]
        self.output_label, self.output_units = ["x", "y"], ["[m]", "[m]"]
        self.x_ub = np.array([10, 10])
        self.x_lb = np.array([-10, -10])

    def f(self, x=np.zeros(3), u=np.zeros(2), t=0):
        dx = np.zeros(self.n)
        dx[0] = u[0]
        dx[1] = u[1]
        return dx

    def device_dynamics(self):
        if not self.stock_model(HolonomicMobileRobot, ("f",)):
            return None
        return _native.DYN_HOLONOMIC, []


class HolonomicMobileRobotwithObstacles(HolonomicMobileRobot):
    """The point robot with forbidden boxes (vehicle_steering.py:336-382; 2D_navigation.py)."""

    def __init__(self):
        super().__init__()
        self.name = "Holonomic Mobile Robot with Obstacles"
        self.x_ub = np.array([10, 10])
        self.x_lb = np.array([-10, -10])
        self.obstacles = [[(2, 2), (4, 10)], [(6, -8), (8, 8)], [(-8, -8), (-1, 8)]]

    def isavalidstate(self, x):
        ans = False
        for i in range(self.n):
            ans = ans or (x[i] < self.x_lb[i])
            ans = ans or (x[i] > self.x_ub[i])
        for obs in self.obstacles:
            on_obs = ((x[0] > obs[0][0]) and (x[1] > obs[0][1]) and (x[0] < obs[1][0]) and (x[1] < obs[1][1]))
            ans = ans or on_obs
        return not ans

    _OBSTACLE_OWNER = None

    def device_obstacles(self):
        return dict(axes=(0, 1), half=(0.0, 0.0), boxes=[[o[0][0], o[0][1], o[1][0], o[1][1]] for o in self.obstacles])


HolonomicMobileRobotwithObstacles._OBSTACLE_OWNER = HolonomicMobileRobotwithObstacles


class KinematicBicyleModel(system.ContinuousDynamicSystem):
    """dx = v cos(theta), dy = v sin(theta), dtheta = v tan(beta) / length."""

    def __init__(self):
        super().__init__(3, 2, 3)
        self.name = "Kinematic Bicyle Model"
        self.state_label, self.state_units = ["x", "y", "theta"], ["[m]", "[m]", "[rad]"]
        self.input_label, self.input_units = ["v", "beta"], ["[m/sec]", "[rad]"]
        self.output_label, self.output_units = ["x", "y", "theta"], ["[m]", "[m]", "[rad]"]
        self.x_ub = np.array([+5, +2, +3.14])
        self.x_lb = np.array([-5, -2, -3.14])
        self.lenght = 1                      # (the reference's spelling: scripts set sys.lenght)
        self.dynamic_domain, self.dynamic_range = True, 10

    def f(self, x=np.zeros(3), u=np.zeros(2), t=0):
        dx = np.zeros(self.n)
        dx[0] = u[0] * np.cos(x[2])
        dx[1] = u[0] * np.sin(x[2])
        dx[2] = u[0] * np.tan(u[1]) * (1. / self.lenght)
        return dx

    def xut2q(self, x, u, t):
        return np.append(x, u[1])

    # ---- device path: closed form PVI_DYN_KINCAR ------------------------------------------------------------------
    def device_dynamics(self):
        if not self.stock_model(KinematicBicyleModel, ("f",)):
            return None
        return _native.DYN_KINCAR, [1. / self.lenght]

    def device_trig(self, x_level):
        return np.cos(x_level[2]), np.sin(x_level[2])

    def device_act_aux(self, input_from_action_id):
        """The per-action heading rate u0 * tan(u1) * (1/length), evaluated here in NumPy exactly as f does."""
        U = np.asarray(input_from_action_id, dtype=float)
        return np.array([U[a, 0] * np.tan(U[a, 1]) * (1. / self.lenght) for a in range(U.shape[0])])


class KinematicCarModel(KinematicBicyleModel):
    """Real-sized car (vehicle_steering.py:717-749)."""

    def __init__(self):
        super().__init__()
        self.width, self.a, self.b = 2.00, 2.00, 3.00
        self.lenght = self.a + self.b
        self.lenght_tire, self.width_tire = 0.40, 0.15
        self.dynamic_domain, self.dynamic_range = True, self.lenght * 2


class KinematicCarModelwithObstacles(KinematicCarModel):
    """The car with non-allowable states: boxes in the (x, y) plane (vehicle_steering.py:973-1021)."""

    def __init__(self):
        super().__init__()
        self.name = "Kinematic Car Model with Obstacles"
        self.obstacles = [[(-10, -1), (-5, 1)], [(-4, 2), (1, 4)], [(12, -1), (17, 1)]]

    def isavalidstate(self, x):
        ans = False
        for i in range(self.n):
            ans = ans or (x[i] < self.x_lb[i])
            ans = ans or (x[i] > self.x_ub[i])
        for obs in self.obstacles:
            on_obs = ((x[0] + self.lenght * 0.5 > obs[0][0]) and (x[1] + self.width * 0.5 > obs[0][1])
                      and (x[0] - self.lenght * 0.5 < obs[1][0]) and (x[1] - self.width * 0.5 < obs[1][1]))
            ans = ans or on_obs
        return not ans

    _OBSTACLE_OWNER = None

    def device_obstacles(self):
        return dict(axes=(0, 1), half=(self.lenght * 0.5, self.width * 0.5),
                    boxes=[[o[0][0], o[0][1], o[1][0], o[1][1]] for o in self.obstacles])


KinematicCarModelwithObstacles._OBSTACLE_OWNER = KinematicCarModelwithObstacles
