"""
Host mirror of pyro/analysis/costfunction.py (CostFunction :19, QuadraticCostFunction :101,
TimeCostFunction :287, QuadraticCostFunctionWithDomainCheck :339, Reachability :421):

    J = int g(x,u,t) dt + h(x(T),T)

QuadraticCostFunction is evaluated in-kernel (device_cost()); the others are served through the
table tier (G built on the host from g(), sweeps on the GPU).
"""
import numpy as np


class CostFunction:
    """INF = out-of-domain cost, EPS = radius of the zero-cost target ball (costfunction.py:31-33)."""

    def __init__(self):
        self.INF = 1e3
        self.EPS = 1e-3

    def h(self, x, t=0):
        raise NotImplementedError

    def g(self, x, u, t=0):
        raise NotImplementedError

    def device_cost(self):
        """dict(Q,R,S,xbar,ubar,EPS,INF,ontarget_check) when libpyrovi can evaluate g/h, else None."""
        return None

    def trajectory_evaluation(self, traj):
        """A copy of `traj` with dJ[i] = g(x[i], u[i], t[i]) and J = its cumulative trapezoidal integral over t, J[0] = 0
        (costfunction.py:56-94)."""
        import copy
        from scipy.integrate import cumulative_trapezoid
        dJ = np.array([self.g(traj.x[i], traj.u[i], traj.t[i]) for i in range(traj.time_steps)], dtype=float)
        out = copy.copy(traj)
        out.J, out.dJ = cumulative_trapezoid(y=dJ, x=traj.t, initial=0), dJ
        return out


def _quad(M, v):
    return float(v @ (np.asarray(M) @ v))


class QuadraticCostFunction(CostFunction):
    """g = dx'Q dx + du'R du, h = dx'S dx, both zero inside ||dx|| < EPS (costfunction.py:151-204)."""

    def __init__(self, n, m):
        super().__init__()
        self.n, self.m = n, m
        self.xbar, self.ubar = np.zeros(n), np.zeros(m)
        self.Q, self.R, self.S = np.eye(n), np.eye(m), np.zeros((n, n))
        self.ontarget_check = True

    @classmethod
    def from_sys(cls, sys):
        cf = cls(sys.n, sys.m)
        cf.xbar, cf.ubar = sys.xbar, sys.ubar          # shared references, as in the reference (:144-145)
        return cf

    def _on_target(self, dx):
        return self.ontarget_check and np.linalg.norm(dx) < self.EPS

    def h(self, x, t=0):
        dx = np.asarray(x, dtype=float) - self.xbar
        return 0 if self._on_target(dx) else _quad(self.S, dx)

    def g(self, x, u, t=0):
        dx = np.asarray(x, dtype=float) - self.xbar
        if self._on_target(dx):
            return 0
        du = np.asarray(u, dtype=float) - self.ubar
        return _quad(self.Q, dx) + _quad(self.R, du)

    def device_cost(self):
        if type(self) is not QuadraticCostFunction:     # subclasses change g/h -> table tier
            return None
        return dict(Q=np.array(self.Q, dtype=float), R=np.array(self.R, dtype=float),
                    S=np.array(self.S, dtype=float), xbar=np.array(self.xbar, dtype=float),
                    ubar=np.array(self.ubar, dtype=float), EPS=float(self.EPS), INF=float(self.INF),
                    ontarget_check=bool(self.ontarget_check))


class TimeCostFunction(CostFunction):
    """g = 1 outside the target ball, h = 0: minimum time (costfunction.py:287-334)."""

    def __init__(self, xbar):
        super().__init__()
        self.xbar = xbar
        self.ontarget_check = True

    def h(self, x, t=0):
        return 0

    def g(self, x, u, t=0):
        if self.ontarget_check and np.linalg.norm(np.asarray(x, dtype=float) - self.xbar) < self.EPS:
            return 0
        return 1

    def device_cost(self):
        """Parameters of the in-kernel evaluation (PVI_COST_TIME)."""
        if type(self) is not TimeCostFunction:          # subclasses change g/h -> table tier
            return None
        return dict(kind="time", xbar=np.asarray(self.xbar, dtype=float), EPS=self.EPS, INF=self.INF,
                    ontarget_check=self.ontarget_check)


class QuadraticCostFunctionWithDomainCheck(QuadraticCostFunction):
    """Quadratic cost that returns INF for states the system rejects (costfunction.py:339-416).
    The on-target zeroing is applied last, as in the reference."""

    def __init__(self, n, m, isavalidstate):
        super().__init__(n, m)
        self.isavalidstate = isavalidstate

    @classmethod
    def from_sys(cls, sys):
        cf = cls(sys.n, sys.m, sys.isavalidstate)
        cf.xbar, cf.ubar = sys.xbar, sys.ubar
        return cf

    def device_cost(self):
        """In-kernel (PVI_COST_QUADRATIC_DOMAIN) when the validity test is a system's own bound isavalidstate: the
        caller (DynamicProgramming._make_engine) checks that it is the system of the grid."""
        if type(self) is not QuadraticCostFunctionWithDomainCheck or getattr(self.isavalidstate, "__self__", None) is None:
            return None
        d = dict(
            Q=np.array(self.Q, dtype=float), R=np.array(self.R, dtype=float), S=np.array(self.S, dtype=float),
            xbar=np.array(self.xbar, dtype=float), ubar=np.array(self.ubar, dtype=float), EPS=float(self.EPS),
            INF=float(self.INF), ontarget_check=bool(self.ontarget_check))
        d["kind"] = "quadratic_domain"
        d["validity_of"] = self.isavalidstate.__self__
        return d

    def h(self, x, t=0):
        dx = np.asarray(x, dtype=float) - self.xbar
        if self._on_target(dx):
            return 0
        return _quad(self.S, dx) if self.isavalidstate(x) else self.INF

    def g(self, x, u, t=0):
        dx = np.asarray(x, dtype=float) - self.xbar
        if self._on_target(dx):
            return 0
        if not self.isavalidstate(x):
            return self.INF
        du = np.asarray(u, dtype=float) - self.ubar
        return _quad(self.Q, dx) + _quad(self.R, du)


class Reachability(CostFunction):
    """h = 0 on the target set else INF; g = 0 on valid states else INF (costfunction.py:421-481)."""

    def __init__(self, isavalidestate, xbar=None, isontarget=None):
        super().__init__()
        self.INF, self.EPS = 1e4, 0.2
        self.isavalidestate = isavalidestate
        if isontarget is None:
            self.isontarget, self.xbar = self.norm_test, xbar
        else:
            self.isontarget = isontarget

    def norm_test(self, x, t=0):
        return np.linalg.norm(np.asarray(x, dtype=float) - self.xbar) < self.EPS

    def device_cost(self):
        """In-kernel (PVI_COST_REACHABILITY) with the default target test and a system's own bound isavalidstate
        (DynamicProgramming._make_engine checks that it is the system of the grid)."""
        if type(self) is not Reachability or getattr(self.isontarget, "__func__", None) is not Reachability.norm_test:
            return None
        if getattr(self.isavalidestate, "__self__", None) is None or self.xbar is None:
            return None
        return dict(kind="reachability", xbar=np.array(self.xbar, dtype=float), EPS=float(self.EPS), INF=float(self.INF),
                    ontarget_check=False, validity_of=self.isavalidestate.__self__)

    def h(self, x, t=0):
        return 0 if self.isontarget(x, t) else self.INF

    def g(self, x, u, t=0):
        return 0 if self.isavalidestate(x) else self.INF
