"""
Host mirror of the part of pyro/dynamic/system.py the value-iteration path touches
(reference: ContinuousDynamicSystem, system.py:21-333): dimensions, labels, box bounds,
nominal point, `f` contract, box validity tests and the Euler step.

Additions for the device path (not in the reference):
  device_dynamics() -> None | (dynamics_id, params, trig_tables_fn)   in-kernel closed form
  f_batch(X, U)     -> dX for B states at once on the GPU when device_dynamics() exists
"""
import numpy as np


class ContinuousDynamicSystem:
    """dx = f(x, u, t) with x in R^n, u in R^m, y in R^p  (reference system.py:21)."""

    def __init__(self, n=1, m=1, p=1):
        self.n, self.m, self.p = n, m, p
        self.name = "ContinuousDynamicSystem"
        self.state_label = ["State %d" % i for i in range(n)]
        self.input_label = ["Input %d" % i for i in range(m)]
        self.output_label = ["Output %d" % i for i in range(p)]
        self.state_units = [""] * n
        self.input_units = [""] * m
        self.output_units = [""] * p
        # default domain (system.py:90-93) and nominal values (:96-98)
        self.x_ub = np.full(n, 10.0)
        self.x_lb = np.full(n, -10.0)
        self.u_ub = np.full(m, 1.0)
        self.u_lb = np.full(m, -1.0)
        self.xbar = np.zeros(n)
        self.ubar = np.zeros(m)
        self.tbar = 0
        self.x0 = np.zeros(n)
        self.traj = None
        self._xbar_init, self._ubar_init = self.xbar, self.ubar      # (what the default cost function is built around)

    # the cost function the reference attaches in __init__ (system.py:121), built on first use -- around the xbar / ubar ARRAYS
    # the system had at the end of ContinuousDynamicSystem.__init__, as there (costfunction.py:139-147 keeps the objects: a
    # subclass or a script that REBINDS sys.xbar / sys.ubar later does not move the default cost's target, an in-place edit does)
    @property
    def cost_function(self):
        if getattr(self, "_cost_function", None) is None:
            from pyro_amd.analysis import costfunction
            cf = costfunction.QuadraticCostFunction.from_sys(self)
            cf.xbar, cf.ubar = self.__dict__.get("_xbar_init", cf.xbar), self.__dict__.get("_ubar_init", cf.ubar)
            self._cost_function = cf
        return self._cost_function

    @cost_function.setter
    def cost_function(self, cf):
        self._cost_function = cf

    # ---- to be provided by subclasses ---------------------------------------------------------
    def f(self, x, u, t=0):
        """State derivative (system.py:124-145)."""
        raise NotImplementedError

    def h(self, x, u, t=0):
        """Output, default y = x (system.py:152-170)."""
        return x

    def t2u(self, t):
        """Open-loop input signal, default constant ubar (system.py:173-191)."""
        return self.ubar

    # ---- domain tests (system.py:198-215): inclusive box; NaN compares false -> valid ---------
    def isavalidstate(self, x):
        x = np.asarray(x)
        return not bool(np.any(x < self.x_lb) or np.any(x > self.x_ub))

    def isavalidinput(self, x, u):
        u = np.asarray(u)
        return not bool(np.any(u < self.u_lb) or np.any(u > self.u_ub))

    # ---- helpers ---------------------------------------------------------------------------------
    def fsim(self, x, t=0):
        """f with the internal input signal (system.py:295-312)."""
        return self.f(x, self.t2u(t), t)

    def x_next(self, x, u, t=0, dt=0.1, steps=1):
        """Explicit Euler, `steps` times (system.py:315-333)."""
        x = np.asarray(x, dtype=float)
        for _ in range(steps):
            x = self.f(x, u, t) * dt + x
        return x

    # ---- after a solve: trajectories (system.py:392-470; pyro_amd/analysis/simulation.py) -----------------
    def compute_trajectory(self, tf=10, n=10001, solver="solve_ivt", **solver_args):
        """Time evolution from self.x0 under the input signal t2u; kept in self.traj (system.py:392-405)."""
        from pyro_amd.analysis import simulation
        self.traj = simulation.Simulator(self, tf, n, solver).compute(**solver_args)
        return self.traj

    def plot_trajectory(self, plot="x", **kwargs):
        """(system.py:408-422; computes the default trajectory when there is none)"""
        from pyro_amd.analysis import simulation
        if self.traj is None:
            self.compute_trajectory()
        return simulation.plot_trajectory(self, self.traj, plot, **kwargs)

    def plot_phase_plane_trajectory(self, x_axis=0, y_axis=1):
        """(system.py:446-459)"""
        from pyro_amd.analysis import simulation
        if self.traj is None:
            self.compute_trajectory()
        return simulation.plot_phase_plane_trajectory(self, self.traj, x_axis, y_axis)

    def animate_simulation(self, **kwargs):
        """The reference draws the system's kinematic sketch along self.traj (system.py:521-560, graphical.Animator).  Graphics
        are outside this build's scope: the trajectory is computed like there, nothing is drawn, and that is said."""
        if self.traj is None:
            self.compute_trajectory()
        print("pyro_amd: animate_simulation() -- animations are not part of this build; the trajectory is in .traj "
              "(%d points over %.3g s)" % (self.traj.time_steps, self.traj.time_final))
        return None

    def show(self, q=None, **kwargs):
        """(system.py:484-497: a still of the kinematic sketch) -- not part of this build, see animate_simulation."""
        print("pyro_amd: show() -- kinematic sketches are not part of this build")
        return None

    # ---- device path -------------------------------------------------------------------------------
    def device_dynamics(self):
        """(dynamics_id, params) when libpyrovi can evaluate f in-kernel, else None."""
        return None

    def device_trig(self, x_level):
        """Host-side sin/cos tables over the grid levels consumed by the kernels."""
        return ()

    def stock_model(self, owner, names):
        """True when every method in `names` is the one class `owner` resolves: no subclass and no instance
        override.  The closed-form kernels hard-code `owner`'s model, so anything else must not use them."""
        for nm in names:
            if nm in self.__dict__ or getattr(type(self), nm, None) is not getattr(owner, nm, None):
                return False
        return True

    def f_batch(self, X, U):
        """dX[b] = f(X[b], U[b]); on the GPU for systems with a closed-form kernel (per-node tables only cover
        the grid nodes), otherwise the plain loop over self.f."""
        from pyro_amd.planning.discretizer import device_dynamics_of
        from pyro_amd import _native
        dd = device_dynamics_of(self)
        X, U = np.atleast_2d(X), np.atleast_2d(U)
        if dd is not None and dd[0] in _native.CLOSED_FORM_IDS:
            return _native.eval_f(dd[0], dd[1], X, U)
        return np.array([self.f(X[i], U[i]) for i in range(X.shape[0])])
