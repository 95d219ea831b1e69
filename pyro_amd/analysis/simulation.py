"""
The callers AFTER the value-iteration path: what the reference's scripts do with the policy a solve returns --
`cl_sys = ctl + sys; cl_sys.compute_trajectory(tf, n, 'euler')` -- i.e. pyro/analysis/simulation.py's Trajectory (:16-90),
Simulator (:208-384) and CLosedLoopSimulator (:391-447), restated on the host.  The closed loop of a look-up-table policy with the
Euler solver is the one case that reaches the GPU: it is ONE row of the batched rollout kernel (k_rollout behind
dp.simulate_closed_loop), see ClosedLoopSystem.compute_trajectory in pyro_amd/control/controller.py.

Out of scope (graphics): the reference's animations.  The plotting shortcuts of a system (plot_trajectory,
plot_phase_plane_trajectory) are thin matplotlib wrappers so that the reference's scripts run unmodified.
"""
import numpy as np


class Trajectory:
    """Time series of one simulation (simulation.py:16-90): x [n, sys.n], u [n, sys.m], t [n], dx, y [n, sys.p]; r = the
    closed loop's reference signal; J / dJ = cumulative and instantaneous cost (CostFunction.trajectory_evaluation)."""

    _FIELDS = ("x", "u", "t", "dx", "y", "r", "J", "dJ")

    def __init__(self, x, u, t, dx, y, r=None, J=None, dJ=None):
        self.x, self.u, self.t, self.dx, self.y, self.r, self.J, self.dJ = x, u, t, dx, y, r, J, dJ
        self._compute_size()

    def _compute_size(self):
        self.time_final = float(self.t.max())
        self.time_steps = int(self.t.size)
        self.n, self.m = self.x.shape[1], self.u.shape[1]
        self.ubar = np.zeros(self.m)

    def _asdict(self):
        return {k: getattr(self, k) for k in self._FIELDS}

    def save(self, name="trajectory.npy"):
        """One .npy holding the dict of arrays (simulation.py:56-66; `load` reads it back)."""
        np.save(name, np.array({k: v for k, v in self._asdict().items() if v is not None}, dtype=object), allow_pickle=True)

    @classmethod
    def load(cls, name):
        d = np.load(name, allow_pickle=True).item()
        return cls(**{k: d.get(k) for k in cls._FIELDS})

    def copy(self):
        return Trajectory(**{k: (None if v is None else np.array(v, copy=True)) for k, v in self._asdict().items()})


class Simulator:
    """Time integration of dx = sys.fsim(x, t) from sys.x0 (simulation.py:208-384).  solver: 'solve_ivt' (scipy solve_ivp; the
    reference's spelling), 'euler' (explicit, n points over [0, tf]: x[i+1] = x[i] + f(x[i], u(t[i]), t[i]) dt), 'odeint'."""

    def __init__(self, sys, tf=10, n=10001, solver="solve_ivt"):
        self.cds, self.t0, self.tf, self.n, self.solver = sys, 0, tf, n, solver
        self.x0 = np.asarray(sys.x0, dtype=float)
        self.cf = sys.cost_function
        if self.x0.size != sys.n:
            raise ValueError("Number of elements in x0 must be equal to number of states")

    def _signals(self, t, x):
        """u, dx, y along a state history (the loop at the end of every branch of simulation.py:246-352)."""
        s = self.cds
        u, dx, y = np.zeros((t.size, s.m)), np.zeros((t.size, s.n)), np.zeros((t.size, s.p))
        for i in range(t.size):
            u[i] = s.t2u(t[i])
            dx[i] = s.f(x[i], u[i], t[i])
            y[i] = s.h(x[i], u[i], t[i])
        return u, dx, y

    def compute(self, **solver_args):
        s = self.cds
        if self.solver == "solve_ivt":
            from scipy.integrate import solve_ivp
            t_eval = None if self.n is None else np.linspace(self.t0, self.tf, int(self.n))
            sol = solve_ivp(lambda t, y: s.fsim(y, t), t_span=[self.t0, self.tf], y0=self.x0, t_eval=t_eval, **solver_args)
            t, x = sol.t, sol.y.T
            u, dx, y = self._signals(t, x)
        elif self.solver == "euler":
            npts = 10001 if self.n is None else int(self.n)
            t = np.linspace(self.t0, self.tf, npts)
            x, dx = np.zeros((npts, s.n)), np.zeros((npts, s.n))
            u, y = np.zeros((npts, s.m)), np.zeros((npts, s.p))
            x[0] = self.x0
            dt = (self.tf + 0.0 - self.t0) / (npts - 1)
            for i in range(npts):
                u[i] = s.t2u(t[i])
                if i + 1 < npts:                    # (the last point has no derivative: it stays zero, as in the reference)
                    dx[i] = s.f(x[i], u[i], t[i])
                    x[i + 1] = dx[i] * dt + x[i]
                y[i] = s.h(x[i], u[i], t[i])
        elif self.solver == "odeint":
            from scipy.integrate import odeint
            npts = 10001 if self.n is None else int(self.n)
            t = np.linspace(self.t0, self.tf, npts)
            x = odeint(s.fsim, self.x0, t)
            u, dx, y = self._signals(t, x)
        else:
            print("Check the solver argument: self.solver ==???")
            raise NotImplementedError
        traj = Trajectory(x=x, u=u, t=t, dx=dx, y=y)
        if self.cf is not None:
            traj = self.cf.trajectory_evaluation(traj)
        return traj


class CLosedLoopSimulator(Simulator):
    """Simulator of a ClosedLoopSystem that also returns the PLANT's inputs (simulation.py:391-447): the combined system's input
    is the reference r; u[i] = controller.c(y[i], r[i], t[i]); the cost is the plant's."""

    def __init__(self, cl_sys, tf=10, n=10001, solver="ode"):
        super().__init__(cl_sys, tf, n, solver)
        self.plant_cf = cl_sys.plant.cost_function

    def _compute_control_inputs(self, traj):
        c = self.cds.controller
        return np.array([np.atleast_1d(c.c(traj.y[i], traj.u[i], traj.t[i])) for i in range(traj.t.size)], dtype=float).reshape(
            traj.t.size, self.cds.plant.m)

    def compute(self):
        traj = Simulator.compute(self)
        return finish_closed_loop(self.cds, traj, self._compute_control_inputs(traj))


def finish_closed_loop(cl_sys, traj, u):
    """The closed-loop trajectory from the combined system's (simulation.py:407-423): its input becomes r, the plant's inputs
    u, the cost the plant's."""
    out = Trajectory(x=traj.x, u=u, t=traj.t, dx=traj.dx, y=traj.y, r=traj.u.copy())
    cf = cl_sys.plant.cost_function
    return cf.trajectory_evaluation(out) if cf is not None else out


# ---------------------------------------------------------------------------------------------------------------- plots
def plot_trajectory(sys, traj, plot="x", show=True):
    """Time plots of a trajectory: any of 'x', 'u', 'y', 'j' in `plot`, one axis per signal (graphical.py's TrajectoryPlotter,
    reduced to what the DP demo scripts call)."""
    import matplotlib.pyplot as plt
    rows = []
    if "x" in plot:
        rows += [(traj.x[:, i], "%s %s" % (sys.state_label[i], sys.state_units[i])) for i in range(traj.x.shape[1])]
    if "u" in plot:
        lab = getattr(sys, "plant", sys)
        rows += [(traj.u[:, i], "%s %s" % (lab.input_label[i], lab.input_units[i])) for i in range(traj.u.shape[1])]
    if "y" in plot:
        rows += [(traj.y[:, i], "%s %s" % (sys.output_label[i], sys.output_units[i])) for i in range(traj.y.shape[1])]
    if "j" in plot and traj.J is not None:
        rows += [(traj.dJ, "dJ"), (traj.J, "J")]
    fig, axes = plt.subplots(max(len(rows), 1), 1, sharex=True, figsize=(4, 3), dpi=200, squeeze=False)
    for ax, (v, label) in zip(axes[:, 0], rows):
        ax.plot(traj.t, v, "b")
        ax.set_ylabel(label, fontsize=5)
        ax.grid(True)
        ax.tick_params(labelsize=5)
    axes[-1, 0].set_xlabel("Time [sec]", fontsize=5)
    fig.canvas.manager.set_window_title("Trajectory for " + sys.name) if getattr(fig.canvas, "manager", None) else None
    fig.tight_layout()
    if show:
        plt.show()
    return fig, axes


def plot_phase_plane_trajectory(sys, traj, x_axis=0, y_axis=1, show=True):
    """The trajectory in the (x_axis, y_axis) plane of the state space, start and end marked."""
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(figsize=(4, 3), dpi=200)
    ax.plot(traj.x[:, x_axis], traj.x[:, y_axis], "b-", linewidth=0.8)
    ax.plot([traj.x[0, x_axis]], [traj.x[0, y_axis]], "o")
    ax.plot([traj.x[-1, x_axis]], [traj.x[-1, y_axis]], "s")
    ax.set_xlabel("%s %s" % (sys.state_label[x_axis], sys.state_units[x_axis]), fontsize=5)
    ax.set_ylabel("%s %s" % (sys.state_label[y_axis], sys.state_units[y_axis]), fontsize=5)
    ax.set_xlim(sys.x_lb[x_axis], sys.x_ub[x_axis])
    ax.set_ylim(sys.x_lb[y_axis], sys.x_ub[y_axis])
    ax.grid(True)
    ax.tick_params(labelsize=5)
    fig.tight_layout()
    if show:
        plt.show()
    return fig, ax
