"""
Host mirror of pyro/planning/dynamicprogramming.py: DynamicProgramming (:115),
DynamicProgrammingWithLookUpTable (:505), LookUpTableController (:27).  Same constructor
arguments, attributes and method names; the Bellman backups run on the GPU through libpyrovi
(include/pyrovi.h).  There is no CPU path: without the library / a HIP device the constructor
raises.

Two device tiers behind one API:
  fused  : system with device_dynamics() + QuadraticCostFunction -> f, Euler, validity, cost,
           interpolation and min/argmin in one kernel, no tables at all;
  table  : any other sys / cf -> x_next_table and G are built on the host exactly like the
           reference (Python loops over sys.f / cf.g), uploaded once; sweeps run on the GPU.

Extra keyword arguments (not in the reference): dtype ('float64' default, 'float32'), device, comm.
J, pi, J_next are NumPy views of device state, fetched on access.

Multi-GPU: `comm=` (pyro_amd.parallel.RcclComm(rank, world, comm_id) | TransportComm | TorchDistComm) shards the grid
by axis-0 slabs over the ranks of the communicator, one process per GPU; every method keeps its meaning and every rank
calls it (they are collective): compute_steps / solve_bellman_equation stop at the same sweep everywhere, J / pi / J_next
are the WHOLE grid on every rank (gathered on access), clean_infeasible_set / get_lookup_table_controller / save_latest
work on those.
"""
import time

import numpy as np

from pyro_amd import _native
from pyro_amd.control import controller
from pyro_amd.planning.discretizer import device_cost_of, device_dynamics_of


class LookUpTableController(controller.StaticController):
    """State feedback u = interp(pi)(x): per input axis a linear interpolant of the INPUT values
    selected by pi, 0 outside the grid (dynamicprogramming.py:27-107)."""

    def __init__(self, grid_sys, pi):
        if grid_sys.nodes_n != pi.size:
            raise ValueError("Grid size does not match optimal action table size")
        super().__init__(1, grid_sys.sys.m, grid_sys.sys.n)
        self.grid_sys, self.pi = grid_sys, pi
        self.name = "Tabular Controller"
        self.interpol_method = ["linear"] * self.m
        self.compute_interpol_functions()

    def compute_interpol_functions(self):
        self.u_interpol = [
            self.grid_sys.compute_interpolation_function(self.grid_sys.get_input_from_policy(self.pi, k),
                                                         self.interpol_method[k], bounds_error=False, fill_value=0)
            for k in range(self.m)]

    def lookup_table_selection(self, x):
        return np.array([float(np.ravel(self.u_interpol[k](x))[0]) for k in range(self.m)])

    def c(self, y, r, t=0):
        return self.lookup_table_selection(y)


class DynamicProgramming:
    """Value iteration on a grid (dynamicprogramming.py:115-500).

    For box-only validity the reference's cell-by-cell base class and its look-up-table subclass
    compute the same recursion; both map to the same fused kernel here."""

    HARD_INF = True                 # base class: an invalid action / next state costs exactly INF (:225-233)
    HISTORY_MAX_BYTES = 1 << 30     # save_time_history is dropped beyond this (J+pi per sweep)
    BATCH = 256                     # sweeps enqueued per host round trip when no history is kept
    INTERPOLATION = "linear"        # interpolant of J_k between the nodes (discretizer.py:570-587)
    F32_FEEDBACK = False            # float32 on 4-D grids: error-feedback storage of J (PVI_FLAG_F32_FEEDBACK, include/pyrovi.h):
                                    # every node keeps the rounding residual of its stored value, so the float32 iterates stay
                                    # within ~2e-7 of the float64 ones over thousands of sweeps (plain float32 storage: up to
                                    # 1.5e-5 mid-solve on BASELINE configs[2]; INTEGRATION.md, accuracy contract)

    INTERNAL_ORDER = "reference"    # "swapped": the cart-pole in float32 with q = (theta, x) inside the engine -- the lanes of the 4-D
                                    # window sweep then run along the axis the displacement does not depend on; J and pi keep the
                                    # reference's node order at this surface (pyro_amd/planning/permuted.py; opt-in, round 5)

    def __init__(self, grid_sys, cost_function, final_time=0, dtype="float64", device=0, comm=None, f32_feedback=None,
                 internal_order=None):
        if f32_feedback is not None:
            self.F32_FEEDBACK = bool(f32_feedback)
        if internal_order is not None:
            if internal_order not in ("reference", "swapped"):
                raise ValueError("internal_order: 'reference' or 'swapped'")
            self.INTERNAL_ORDER = internal_order
        self.grid_sys, self.sys = grid_sys, grid_sys.sys
        self.cf, self.tf = cost_function, final_time
        self.alpha = 1.0
        self.save_time_history = True
        self.verbose = True
        self.stats_every_sweep = True   # sharded grids: all-reduce the statistics after every sweep (same on all ranks)
        self.t, self.k = self.tf, 0
        self.start_time = time.time()
        self.dtype, self.device, self.comm = np.dtype(dtype), device, comm
        self._make_engine()
        self.interpol_method = self.INTERPOLATION
        self.evaluate_terminal_cost()
        self.t_list, self.J_list, self.pi_list = [], [], []
        if self._history_ok():
            self.t_list, self.J_list, self.pi_list = [self.tf], [self.J], [self.pi]

    # ------------------------------------------------------------------ device engine
    # interpolants of J_k the sweeps implement (discretizer.py:570-587 hands dp.interpol_method to RegularGridInterpolator
    # every sweep; 'bicubic' is this build's name for the RectBivariateSpline subclass, dynamicprogramming.py:578-614)
    _INTERPOLATIONS = ("linear", "slinear", "nearest", "cubic", "cubic_legacy", "bicubic")
    _SAME_ENGINE = {"slinear": "linear", "cubic_legacy": "cubic"}          # (methods the same engine serves)

    @property
    def interpol_method(self):
        return self.__dict__.get("_interpol_method", "linear")

    @interpol_method.setter
    def interpol_method(self, value):
        """The reference passes dp.interpol_method to the interpolant of every sweep (dynamicprogramming.py:186-189,
        discretizer.py:570-587); here the interpolation is compiled into the engine, so:
          'linear'  -- every tier (default);
          'slinear' -- RegularGridInterpolator's order-1 spline: the SAME interpolant as 'linear' evaluated by another code path of
                       scipy (the reference's own 12-sweep solves with the two differ by 1.8e-15 of max J, identical policies:
                       tests/golden/slinear_*.npz).  Served by the linear sweeps of every tier; no engine is rebuilt;
          'nearest' -- RegularGridInterpolator(method='nearest'): the TABLE tier implements it (the interval and fraction of
                       every cell are fixed when the tables are packed).  Assigning it rebuilds the engine on the table
                       tier from the reference's look-up tables and carries the current cost-to-go (J, J_next, pi, k) over:
                       the new interpolant applies from the next sweep on, as in the reference; a sharded engine raises;
          'cubic', 'cubic_legacy' -- 2-D grids: RegularGridInterpolator's order-3 spline, i.e. the interpolating tensor spline with
                       not-a-knot ends (the spline of RectBivariateSpline(kx=ky=3): SciPy's 'cubic_legacy' agrees with it to 3e-15)
                       inside the grid box and 0 outside.  Served by the spline sweep of the TABLE tier (refit every sweep) with the
                       in-box mask of x_next as its validity table; the engine is rebuilt like for 'nearest'.  SciPy >= 1.13 fits
                       'cubic' with an iterative solver at its default tolerance and is itself some 1e-5 of max|J| away from the
                       spline it approximates (tests/golden/cubic_pendulum.npz records both): this build returns the exact fit
                       for either name.  Grids of other dimensions raise;
          'bicubic' -- only as the class DynamicProgramming2DRectBivariateSpline (the same spline, CLAMPED outside the box);
        anything else ('quintic', 'pchip') raises instead of silently computing with another interpolant."""
        if value not in self._INTERPOLATIONS:
            raise NotImplementedError("interpol_method %r: the GPU sweeps implement %s" % (value, ", ".join(self._INTERPOLATIONS)))
        if self._SAME_ENGINE.get(value, value) == "cubic":
            if self.sys.n != 2:
                raise NotImplementedError("interpol_method %r: the spline sweep is 2-D (this grid has %d axes)" % (value, self.sys.n))
            if self.INTERNAL_ORDER == "swapped" or self.F32_FEEDBACK:
                raise NotImplementedError("interpol_method %r runs on the table tier: no internal_order / f32_feedback" % (value,))
        if value == "bicubic" or self.INTERPOLATION == "bicubic":
            if "_p" in self.__dict__ and value != self.INTERPOLATION:
                raise NotImplementedError("interpol_method %r on a %s engine: use %s" % (
                    value, self.INTERPOLATION, "DynamicProgramming2DRectBivariateSpline" if value == "bicubic"
                    else "DynamicProgrammingWithLookUpTable"))
            self.__dict__["_interpol_method"] = value
            return
        old = self.__dict__.get("_interpol_method", "linear")
        self.__dict__["_interpol_method"] = value
        same = self._SAME_ENGINE
        if "_p" in self.__dict__ and same.get(value, value) != same.get(old, old):
            if self.comm is not None or not hasattr(self, "_rebuild_engine"):
                self.__dict__["_interpol_method"] = old
                raise NotImplementedError("interpol_method %r on this engine (sharded grids and policy evaluation keep the "
                                          "interpolant they were built with)" % value)
            self._rebuild_engine()

    def _feedback_applies(self):
        """Error-feedback storage exists where a float32 production sweep exists: 4-D grids (k_sweep_lean4fb), 2-D grids of
        one-input mechanical systems (k_sweep_leanfb: the pendulum family and the per-node-table tier, e.g. MountainCar) and the
        explicit systems (k_sweep3_fast).  The library itself refuses the flag on a handle that does not take such a sweep
        (PVI_EINVAL)."""
        if self.dtype != np.float32:
            return False
        # Only the 4-D form has run on hardware.  The 2-D and explicit-system forms were written while no MI355X was reachable
        # (round 5): until tests/test_gpu_zz_unproven.py has passed they are admitted with _native.overrides(UNPROVEN="1") only --
        # the library applies the same gate (pvi_create: PVI_EINVAL).
        unproven = _native.override_value("UNPROVEN") == "1"
        mech = getattr(self.sys, "dof", None)
        if mech is not None:        # mechanical systems: the LDS-window sweeps
            return self.sys.n == 4 or (unproven and self.sys.n == 2 and self.sys.m == 1)
        return unproven             # explicit systems: k_sweep3_fast (the library refuses where that sweep does not apply)

    def _make_engine(self):
        self._host = {}             # cached downloads: 'J', 'pi', 'J_next'
        self._dirty = False         # host J newer than the device copy
        if self.INTERNAL_ORDER == "swapped" and (self.comm is not None or self.INTERPOLATION != "linear"):
            raise NotImplementedError("internal_order='swapped': one GPU, linear interpolation")
        if self.comm is not None:
            if self.INTERPOLATION != "linear":
                raise NotImplementedError("the spline fit couples every row of the grid: single-GPU only")
            if self.F32_FEEDBACK and not self._feedback_applies():
                raise NotImplementedError("f32_feedback is the float32 storage mode of the 4-D LDS-window sweep (2-D grids with one input and "
                                          "explicit systems: written, not yet run on hardware -- _native.overrides(UNPROVEN='1') admits "
                                          "them; dtype %s, n = %d, m = %d)" % (self.dtype, self.sys.n, self.sys.m))
            self._p = self.comm.engine(self)        # (every sharded engine carries the flag to the pieces of its slabs)
            self.tier = self._p.tier
            return
        dd = device_dynamics_of(self.sys)
        cost = device_cost_of(self.cf, self.sys)
        self.tier = "fused" if (dd is not None and cost is not None) else "table"
        if self.INTERPOLATION != "linear" and dd is not None and dd[0] != _native.DYN_PENDULUM:
            self.tier = "table"         # the spline sweep has in-kernel dynamics for the pendulum family only
        nearest = self.__dict__.get("_interpol_method", "linear") == "nearest"
        method = self.__dict__.get("_interpol_method", "linear")
        cubic = self._SAME_ENGINE.get(method, method) == "cubic"
        if nearest or cubic:
            self.tier = "table"         # nearest-neighbour interpolation: packed into the table tier's records;
                                        # 'cubic': the spline sweep over the raw tables with the in-box mask
        if self.tier == "fused":
            # (base class: an invalid cell costs exactly INF; the same as INF + alpha*0 unless the system rejects
            #  states inside the grid box, i.e. obstacles)
            if self.F32_FEEDBACK and not self._feedback_applies():
                raise NotImplementedError("f32_feedback is the float32 storage mode of the 4-D LDS-window sweep (2-D grids with one input and "
                                          "explicit systems: written, not yet run on hardware -- _native.overrides(UNPROVEN='1') admits "
                                          "them; dtype %s, n = %d, m = %d)" % (self.dtype, self.sys.n, self.sys.m))
            flags = (_native.FLAG_HARD_INF if self.HARD_INF else 0) | (_native.FLAG_F32_FEEDBACK if self.F32_FEEDBACK else 0)
            if self.INTERNAL_ORDER == "swapped":
                from pyro_amd.planning import permuted
                if not permuted.swap_applies(self.sys, dd, self.dtype):
                    raise NotImplementedError("internal_order='swapped' is the stock cart-pole's closed form in float32 (%s, %s)"
                                              % (type(self.sys).__name__, self.dtype))
                kw = self.grid_sys._problem_kwargs(cost, self.dtype, device=self.device, flags=flags)
                self._p = permuted.SwappedProblem(_native.Problem(**permuted.swap_problem_kwargs(kw)), self.grid_sys.x_grid_dim)
            else:
                self._p = self.grid_sys._device_problem(cost=cost, dtype=self.dtype, device=self.device, flags=flags)
        else:
            if self.INTERNAL_ORDER == "swapped":
                raise NotImplementedError("internal_order='swapped': the fused tier only (this problem runs on the table tier)")
            if self.F32_FEEDBACK:
                raise NotImplementedError("f32_feedback: the fused tier only (this problem runs on the table tier)")
            g, s = self.grid_sys, self.sys
            self._p = _native.Problem(g.x_level, g.u_level, s.x_lb, s.x_ub, s.u_lb, s.u_ub, g.dt, dtype=self.dtype,
                                      dynamics_id=_native.DYN_TABLE, cost=None, device=self.device,
                                      table_inf=float(self.cf.INF))
            if self.INTERPOLATION != "linear":
                self._p.set_interpolation(self.INTERPOLATION)       # before the tables: the spline sweep reads them raw
            elif nearest:
                self._p.set_interpolation("nearest")                # before the tables: fractions are snapped when they are packed
            elif cubic:
                self._p.set_interpolation("bicubic")
            ok = (g.action_isok & g.x_next_isok) if self.HARD_INF else None
            if cubic and ok is None:
                # RegularGridInterpolator(bounds_error=False, fill_value=0): a cell outside the box of the grid LEVELS takes
                # J = 0, so Q = G + alpha * 0 = INF exactly (G is INF there: x_next_isok is false outside the box) -- the
                # kernel's validity table does that; inside the box the look-up-table class adds the spline even where G = INF
                xn = g.x_next_table
                ok = np.ones(xn.shape[:2], dtype=bool)
                for d in range(s.n):
                    ok &= ~(xn[:, :, d] < g.x_level[d][0]) & ~(xn[:, :, d] > g.x_level[d][-1])
            self._p.set_tables(g.x_next_table, self._host_cost_table(), ok)
        if self.tier == "fused" and self.INTERPOLATION != "linear":
            self._p.set_interpolation(self.INTERPOLATION)

    @property
    def sharded(self):
        return bool(getattr(self._p, "sharded", False))

    def _cost_rows(self, lo, hi):
        g = self.grid_sys
        X, U = g.state_from_node_id, g.input_from_action_id
        ok = (g.action_isok & g.x_next_isok)[lo:hi]
        G = np.full((hi - lo, g.actions_n), float(self.cf.INF))
        for s, a in zip(*np.nonzero(ok)):
            G[s, a] = self.cf.g(X[lo + s], U[a], self.t) * g.dt
        return G

    def _host_cost_table(self):
        """compute_cost_lookuptable for arbitrary cf.g (dynamicprogramming.py:517-553); large tables are split over
        the host cores like the x_next table."""
        g = self.grid_sys
        g.state_from_node_id, g.input_from_action_id, g.action_isok, g.x_next_isok     # materialise before forking
        from pyro_amd.planning.discretizer import host_parallel_rows
        return np.concatenate(host_parallel_rows(self, "_cost_rows", g.nodes_n, g.nodes_n * g.actions_n))

    # ------------------------------------------------------------------ J / pi live on the device
    def _flush(self):
        if self._dirty:
            self._p.set_J(self._host["J"])
            self._dirty = False

    @property
    def J(self):
        if "J" not in self._host:
            self._host["J"] = self._p.get_J()
        return self._host["J"]

    @J.setter
    def J(self, value):
        value = np.asarray(value, dtype=float)
        if value.size != self.grid_sys.nodes_n:
            raise ValueError("Grid size does not match data")
        self._host["J"] = value
        self._dirty = True

    @property
    def pi(self):
        if "pi" not in self._host:
            self._host["pi"] = self._p.get_pi()
        return self._host["pi"]

    @pi.setter
    def pi(self, value):
        self._host["pi"] = np.asarray(value).astype(int)

    @property
    def J_next(self):
        if "J_next" not in self._host:
            self._host["J_next"] = self._p.get_J(prev=True) if self.k > 0 else self.J
        return self._host["J_next"]

    @J_next.setter
    def J_next(self, value):
        self._host["J_next"] = np.asarray(value, dtype=float)

    def _invalidate(self):
        self._host.clear()
        self._dirty = False

    def _rebuild_engine(self):
        """A new engine for the current interpol_method, carrying the current cost-to-go over: the reference builds its
        interpolant from J_next with the CURRENT method at the start of every sweep (dynamicprogramming.py:186-189), so a
        change of method between sweeps applies from the next sweep on."""
        J, pi = self.J.copy(), self.pi.copy()
        J_next = self.J_next.copy()         # (the fresh handle's second buffer is zero: keep the host copy until the next sweep)
        self._p.close()
        self.__dict__.pop("_G", None)
        self._make_engine()
        self.J = J
        self._host["pi"] = pi
        self._host["J_next"] = J_next

    # ------------------------------------------------------------------ reference API
    def evaluate_terminal_cost(self):
        """J = cf.h(x, tf), pi = 0 (dynamicprogramming.py:159-171)."""
        if self.tier == "fused" or self.sharded:
            self._p.terminal_cost()         # (sharded table tier: every rank evaluates cf.h on its own rows)
            self._invalidate()
        else:
            X = self.grid_sys.state_from_node_id
            self.J = np.array([self.cf.h(X[s], self.tf) for s in range(self.grid_sys.nodes_n)], dtype=float)
            self._flush()

    def _history_ok(self):
        if not self.save_time_history:
            return False
        per_sweep = self.grid_sys.nodes_n * 16
        if per_sweep * (len(self.J_list) + 1) > self.HISTORY_MAX_BYTES:
            if self.verbose:
                print("save_time_history dropped: history would exceed %d bytes" % self.HISTORY_MAX_BYTES)
            self.save_time_history = False
            return False
        return True

    def _run(self, max_sweeps, tol):
        """Enqueue up to max_sweeps backups (stop on delta <= tol when tol >= 0); returns delta."""
        self._flush()
        delta, done = None, 0
        while done < max_sweeps:
            nb = 1 if self._history_ok() else min(self.BATCH, max_sweeps - done)
            if self.sharded:
                # statistics cost a host synchronisation + an all-reduce per sweep on a sharded grid: every sweep when
                # they are tested (tol) or wanted (stats_every_sweep, the reference's behaviour), else only the last of the
                # batch.  The switch decides which collectives run, so it must be the same on every rank -- unlike
                # `verbose`, which only decides whether THIS rank prints them.
                stats, n = self._p.sweep(nb, self.alpha, tol, every=self.stats_every_sweep)
                quiet = n - len(stats)
                self.k += quiet
                self.t -= self.grid_sys.dt * quiet
            else:
                stats, n = self._p.sweep(nb, self.alpha, tol)
            self._invalidate()
            for st in stats:
                self.k += 1
                self.t -= self.grid_sys.dt
                delta = self._report(st)
            done += n
            if self.save_time_history and n:
                self.J_list.append(self.J)
                self.t_list.append(self.t)
                self.pi_list.append(self.pi)
            if n < nb or (tol >= 0 and delta is not None and delta <= tol):
                break
        return delta

    def _report(self, st):
        """finalize_backward_step (dynamicprogramming.py:240-261): print and return delta."""
        if self.verbose:
            print("%d t:%.2f Elasped:%.2f max: %.2f dmax:%.2f dmin:%.2f"
                  % (self.k, self.t, time.time() - self.start_time, st[0], st[1], st[2]))
        return float(st[3])

    # the three-step form of one backup, kept for scripts that drive it by hand (:175-261)
    def initialize_backward_step(self):
        self._pending = True

    def compute_backward_step(self):
        self._flush()
        self._last_stats, n = self._p.sweep(1, self.alpha, -1.0)[:2]
        self._invalidate()
        self.k += 1
        self.t -= self.grid_sys.dt

    def finalize_backward_step(self):
        delta = self._report(self._last_stats[0])
        if self._history_ok():
            self.J_list.append(self.J)
            self.t_list.append(self.t)
            self.pi_list.append(self.pi)
        return delta

    def _animated(self, max_sweeps, tol, animate_cost2go, animate_policy, k):
        """Sweep-by-sweep driver used when a live plot is requested (one host round trip per sweep)."""
        if animate_cost2go:
            self.plot_cost2go()
        if animate_policy:
            self.plot_policy(k)
        delta, done = self.cf.INF, 0
        while done < max_sweeps and (tol < 0 or delta > tol):
            delta = self._run(1, -1.0)
            done += 1
            if animate_cost2go:
                self.update_cost2go_plot()
            if animate_policy:
                self.update_policy_plot()
        return delta

    def compute_steps(self, n=50, animate_cost2go=False, animate_policy=False, k=0):
        print("\nComputing %d backward DP iterations:" % n)
        print("-----------------------------------------")
        if animate_cost2go or animate_policy:
            self._animated(n, -1.0, animate_cost2go, animate_policy, k)
        else:
            self._run(n, -1.0)

    def solve_bellman_equation(self, tol=0.1, animate_cost2go=False, animate_policy=False, k=0):
        """Backups until max|J - J_next| <= tol (dynamicprogramming.py:283-314); delta starts at
        cf.INF, so tol >= INF performs no sweep."""
        print("\nComputing backward DP iterations until dJ<%2.2f:" % tol)
        print("---------------------------------------------------------")
        delta = self.cf.INF
        if animate_cost2go or animate_policy:
            if delta > tol:
                self._animated(1 << 30, tol, animate_cost2go, animate_policy, k)
        else:
            while delta > tol:
                delta = self._run(1 << 30, tol)
        print("\nBellman equation solved!")

    def clean_infeasible_set(self, tol=1):
        """J > INF - tol  ->  J = INF, pi = action nearest to sys.ubar (:322-334)."""
        default_action = self.grid_sys.get_nearest_action_id_from_input(self.sys.ubar)
        J, pi = self.J.copy(), self.pi.copy()
        bad = J > (self.cf.INF - tol)
        J[bad] = self.cf.INF
        pi[bad] = default_action
        self.J, self.pi = J, pi

    def get_lookup_table_controller(self):
        ctl = LookUpTableController(self.grid_sys, self.pi)
        ctl._dp = self      # (lets `ctl + sys` hand an Euler trajectory to the rollout kernel: ClosedLoopSystem.compute_trajectory)
        return ctl

    def simulate_closed_loop(self, X0, tf=10.0, n=10001, device_only=False):
        """(not in the reference) B closed-loop Euler trajectories of the current policy on the GPU: what
        `(dp.get_lookup_table_controller() + sys).compute_trajectory(tf, n, 'euler')` does for one x0
        (controller.py:328-355, simulation.py:298-324).  Returns t [n], X [B,n,sys.n], U [B,n,sys.m]."""
        dt = (tf + 0.0) / (n - 1)
        t = np.linspace(0, tf, n)
        closed_form = (self.tier == "fused" and not self.sharded and self._p.dynamics_id in _native.ROLLOUT_IDS
                       and not getattr(self._p, "swapped", False))    # (a swapped engine: the host loop below, in the reference's order)
        rp = self.sys.device_rollout_params() if (closed_form and hasattr(self.sys, "device_rollout_params")) else ()
        if rp is None:
            closed_form = False                 # e.g. a quarter car with its own terrain: arbitrary Python again
        if closed_form:
            if len(rp):                         # constants of the continuous closed form (terrain, tyre curve)
                self._p.set_rollout_params(rp)
            self._p.set_pi(self.pi)             # the host policy may have been edited (clean_infeasible_set)
            X, U = self._p.rollout(X0, n, dt)
            return t, X, U
        if device_only:
            raise NotImplementedError("no rollout kernel for this engine")
        # systems whose f is arbitrary Python (table tier, per-node tables of generic mechanical systems): the reference's loop --
        # u = ctl.c(x, t), x <- x + f(x, u, t) dt (controller.py:328-355, simulation.py:298-324) -- on the host
        ctl = self.get_lookup_table_controller()
        X0 = np.atleast_2d(np.asarray(X0, dtype=float))
        X = np.empty((X0.shape[0], n, self.sys.n))
        U = np.empty((X0.shape[0], n, self.sys.m))
        for b, x in enumerate(X0):
            for i in range(n):
                u = np.atleast_1d(ctl.c(x, t[i]))
                X[b, i], U[b, i] = x, u
                x = x + np.asarray(self.sys.f(x, u, t[i]), dtype=float) * dt
        return t, X, U

    def save_latest(self, name="test_data"):
        """Writes J_next (not J), as the reference does (:481-485)."""
        np.save(name + "_J_inf", self.J_next)
        np.save(name + "_pi_inf", self.pi.astype(int))

    def load_J_next(self, name="test_data"):
        try:
            self.J_next = np.load(name + "_J_inf" + ".npy")
        except Exception:
            print("Failed to load J_next ")

    # ------------------------------------------------------------------ plots: host pass-through (matplotlib)
    # (dynamicprogramming.py:343-464: thin wrappers over the GridDynamicSystem plots, on downloaded arrays)
    def plot_cost2go(self, jmax=None, i=0, j=1, show=True):
        import matplotlib.pyplot as plt
        jmax = self.cf.INF if jmax is None else jmax
        fig, ax, pcm = self.grid_sys.plot_grid_value(self.J, "Cost-to-go", i, j, jmax, 0)
        self.cost2go_fig = [fig, ax, pcm, ax.text(0.05, 0.05, "", transform=ax.transAxes, fontsize=8), i, j]
        plt.ion()
        if show:
            plt.pause(0.001)

    def update_cost2go_plot(self, show=True):
        import matplotlib.pyplot as plt
        g = self.grid_sys
        Z = g.get_2D_slice_of_grid(g.get_grid_from_array(self.J), self.cost2go_fig[4], self.cost2go_fig[5])
        self.cost2go_fig[2].set_array(np.ravel(Z.T))
        self.cost2go_fig[3].set_text("Optimal cost2go at time = %4.2f" % self.t)
        if show:
            plt.pause(0.001)

    def plot_policy(self, k=0, i=0, j=1, show=True):
        import matplotlib.pyplot as plt
        fig, ax, pcm = self.grid_sys.plot_control_input_from_policy(self.pi, k, i, j)
        self.policy_fig = [fig, ax, pcm, ax.text(0.05, 0.05, "", transform=ax.transAxes, fontsize=8), k, i, j]
        plt.ion()
        if show:
            plt.pause(0.001)

    def update_policy_plot(self, show=True):
        import matplotlib.pyplot as plt
        g = self.grid_sys
        uk = g.get_grid_from_array(g.get_input_from_policy(self.pi, self.policy_fig[4]))
        self.policy_fig[2].set_array(np.ravel(g.get_2D_slice_of_grid(uk, self.policy_fig[5], self.policy_fig[6]).T))
        self.policy_fig[3].set_text("Optimal policy at time = %4.2f" % self.t)
        if show:
            plt.pause(0.001)

    def plot_cost2go_3D(self, jmax=None, i=0, j=1, show=True):
        jmax = self.cf.INF if jmax is None else jmax
        fig, ax, surf = self.grid_sys.plot_grid_value_3D(self.J, None, "Cost-to-go", i, j, jmax, 0)
        self.cost2go_3D_fig = [fig, ax, surf, ax.text2D(0.05, 0.05, "", transform=ax.transAxes, fontsize=8), i, j]

    def _frame(self, n, with_policy):
        self.J, self.t = self.J_list[n], self.t_list[n]
        if with_policy:
            self.pi = self.pi_list[n]
        self.clean_infeasible_set()

    def animate_cost2go(self, i=0, j=1, jmax=None, show=True, save=False, file_name="cost2go_animation"):
        """Replays J_list (needs save_time_history)."""
        import matplotlib.animation as animation
        self._frame(0, False)
        self.plot_cost2go(jmax=jmax, i=i, j=j, show=False)

        def step(n):
            self._frame(n, False)
            self.update_cost2go_plot(show=False)
        self.ani = animation.FuncAnimation(self.cost2go_fig[0], step, len(self.J_list), interval=20)
        if save:
            self.ani.save(file_name + ".gif", writer="imagemagick", fps=30)
        if show:
            self.cost2go_fig[0].show()
        return self.ani

    def animate_policy(self, k=0, i=0, j=1, show=True, save=False, file_name="policy_animation"):
        import matplotlib.animation as animation
        self._frame(1, True)
        self.plot_policy(k=k, i=i, j=j, show=False)

        def step(n):
            self._frame(n + 1, True)
            self.update_policy_plot(show=False)
        self.ani = animation.FuncAnimation(self.policy_fig[0], step, len(self.pi_list) - 1, interval=20)
        if save:
            self.ani.save(file_name + ".gif", writer="imagemagick", fps=30)
        if show:
            self.policy_fig[0].show()
        return self.ani


class DynamicProgrammingWithLookUpTable(DynamicProgramming):
    """dynamicprogramming.py:505-570.  `G` is available as an attribute (built on the GPU for the
    fused tier) but the fused sweep never materialises it."""

    HARD_INF = False                # Q = G + alpha*J_interp even where G = INF (:567)

    @property
    def G(self):
        if "_G" not in self.__dict__:
            self.compute_cost_lookuptable()
        return self.__dict__["_G"]

    def compute_cost_lookuptable(self):
        t0 = time.time()
        print("Computing g(x,u,t) look-up table..  ", end="")
        if self.tier == "fused" and self.sharded:
            p = self.grid_sys._device_problem(cost=device_cost_of(self.cf, self.sys), device=self.device)
            self.__dict__["_G"] = p.build_tables(x_next=False, x_next_isok=False, action_isok=False)[3]
            p.close()
        elif self.tier == "fused":
            self.__dict__["_G"] = self._p.build_tables(x_next=False, x_next_isok=False, action_isok=False)[3]
        else:
            self.__dict__["_G"] = self._host_cost_table()
        print("completed in %4.2f sec" % (time.time() - t0))


class DynamicProgramming2DRectBivariateSpline(DynamicProgrammingWithLookUpTable):
    """dynamicprogramming.py:578-614: the look-up-table recursion with a bicubic spline through J_k
    (RectBivariateSpline kx = ky = 3, refit every sweep) instead of the bilinear interpolant; x_next is
    clamped to the grid box by the spline evaluation, nothing is zero-filled.  2-D grids only."""

    INTERPOLATION = "bicubic"

    def __init__(self, grid_sys, cost_function, final_time=0, dtype="float64", device=0, comm=None, f32_feedback=None):
        if grid_sys.sys.n != 2:
            raise NotImplementedError                   # discretizer.py:599-610
        super().__init__(grid_sys, cost_function, final_time, dtype=dtype, device=device, comm=comm, f32_feedback=f32_feedback)

    @property
    def J_interpol(self):
        """The interpolant of the current J_next as a SciPy object (host side, for inspection / plots)."""
        return self.grid_sys.compute_bivariatespline_2D_interpolation_function(self.J_next, kx=3, ky=3)


class PolicyEvaluator(DynamicProgramming):
    """Cost-to-go of a GIVEN control law u = ctl.c(x, ctl.rbar, t) (dynamicprogramming.py:623-677): the
    same backup with exactly one action per node.  The controller is arbitrary Python, so x_next and G are
    evaluated on the host node by node like the reference (:704-735); the sweeps run on the GPU (table
    tier, A = 1).  Base class: an invalid input / next state costs exactly INF."""

    HARD_INF = True

    def __init__(self, ctl, grid_sys, cost_function, final_time=0, dtype="float64", device=0, comm=None):
        self.ctl = ctl
        super().__init__(grid_sys, cost_function, final_time, dtype=dtype, device=device, comm=comm)

    def _node_tables(self, lo, hi):
        """(x_next [hi-lo, n], G [hi-lo], ok [hi-lo]) of the nodes lo..hi-1 by the reference's loop (dynamicprogramming.py:
        704-735): u = ctl.c(x), x_next = f dt + x, ok = isavalidinput and isavalidstate, G = g dt or INF."""
        g, s = self.grid_sys, self.sys
        X, r = g.state_from_node_id, self.ctl.rbar
        xn, G, ok = np.zeros((hi - lo, s.n)), np.zeros(hi - lo), np.zeros(hi - lo, dtype=bool)
        for k, i in enumerate(range(lo, hi)):
            x = X[i]
            u = self.ctl.c(x, r, self.t)
            x_next = s.f(x, u, self.t) * g.dt + x
            xn[k] = x_next
            ok[k] = s.isavalidinput(x, u) and s.isavalidstate(x_next)
            G[k] = self.cf.g(x, u, self.t) * g.dt if ok[k] else self.cf.INF
        return xn, G, ok

    def _make_engine(self):
        g, s = self.grid_sys, self.sys
        N = g.nodes_n
        if self.comm is not None:
            # Sharded (round 4): every rank evaluates the control law, f and g on ITS rows of axis 0 only -- the reference's
            # O(N) Python loop split over the ranks -- and the one-action tables go to the library's sharded table tier
            # (halo width from the table itself, agreed between the ranks).  x_next_table / G hold this rank's rows.
            if not hasattr(self.comm, "comm_id") and not hasattr(self.comm, "sendrecv"):
                raise NotImplementedError("policy evaluation over a sharded grid: RcclComm or TransportComm (the library's "
                                          "sharded table tier)")
            self.tables_on = "host (this rank's rows)"
            one = [np.zeros(1) for _ in range(s.m)]

            def build(lo, hi):
                xn, G, ok = self._node_tables(lo, hi)
                self.x_next_table, self.G = xn, G
                return xn[:, None, :], G[:, None], (ok[:, None] if self.HARD_INF else None)
            self._shard_tables = dict(u_levels=one, u_lb=np.zeros(s.m), u_ub=np.zeros(s.m), build=build)
            self._p = self.comm.engine(self)
            self.tier = "table"
            self._host = {}
            self._dirty = False
            return
        self.tables_on = "host"
        dd = device_dynamics_of(s)
        cost = device_cost_of(self.cf, s)
        closed = (dd is not None and dd[0] in _native.CLOSED_FORM_IDS and cost is not None and "kind" not in cost)
        ctl_dev = self.ctl.device_controller(s) if (closed and hasattr(self.ctl, "device_controller")) else None
        if closed:
            # u(x) by the library when the control law has a kernel (the reference's ComputedTorqueController), else by
            # the O(N) Python calls of ctl.c; f, isavalidinput / isavalidstate and g in ONE kernel either way (the
            # reference loops over the nodes in Python for all of it, dynamicprogramming.py:704-735)
            p = g._device_problem(cost=cost, device=self.device)
            try:
                if ctl_dev is not None:
                    q = np.asarray(ctl_dev[1], dtype=float)
                    k1, k2 = 2 * q[-2] * q[-1], q[-1] ** 2           # (2 zeta) w0 and w0^2: nonlinear.py:107
                    U, xn, ok, G = p.policy_tables(ctl_dev[0], np.concatenate([q[:-2], [k1, k2]]))
                    self.tables_on = "gpu: controller + dynamics + cost"
                else:
                    X, r = g.state_from_node_id, self.ctl.rbar
                    U0 = np.array([np.atleast_1d(self.ctl.c(X[i], r, self.t)) for i in range(N)], dtype=float)
                    U, xn, ok, G = p.policy_tables(_native.CTL_TABLE, None, U0)
                    self.tables_on = "gpu: dynamics + cost (controller on the host)"
            finally:
                p.close()
            self.U, self.x_next_table, self.G = U, xn, G
        else:
            self.x_next_table, self.G, ok = self._node_tables(0, N)
        one = [np.zeros(1) for _ in range(s.m)]                     # a single placeholder action
        self.tier = "table"
        self._p = _native.Problem(g.x_level, one, s.x_lb, s.x_ub, np.zeros(s.m), np.zeros(s.m), g.dt, dtype=self.dtype,
                                  dynamics_id=_native.DYN_TABLE, cost=None, device=self.device,
                                  table_inf=float(self.cf.INF))
        if self.__dict__.get("_interpol_method", "linear") == "nearest":
            self._p.set_interpolation("nearest")
        self._p.set_tables(self.x_next_table[:, None, :], self.G[:, None], ok[:, None] if self.HARD_INF else None)
        self._host = {}
        self._dirty = False


class PolicyEvaluatorWithLookUpTable(PolicyEvaluator):
    """dynamicprogramming.py:683-753: J = G + alpha * J_interp(x_next_table), G = INF on invalid nodes."""

    HARD_INF = False
