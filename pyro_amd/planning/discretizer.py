"""
Host mirror of pyro/planning/discretizer.py:24-834 (GridDynamicSystem): regular grids over the
state and input boxes, node/action enumeration (C order), index helpers, interpolation objects
and the forward-dynamics look-up tables.

Differences a user can see (DESIGN.md "drop-in notes"):
  * the O(N) / O(N*A) arrays (state_from_node_id, x_next_table, ...) are built on first access
    instead of in __init__, so a 101^4 grid does not allocate 70 GB up front;
  * x_next_table / x_next_isok / action_isok come from the GPU (pvi_build_tables) when the
    system has in-kernel dynamics, and from the reference's node-by-node loop over sys.f otherwise.
"""
import time

import numpy as np
from scipy.interpolate import RectBivariateSpline, RegularGridInterpolator

from pyro_amd import _native


def _null_cost(n, m):
    return dict(Q=np.eye(n), R=np.eye(m), S=np.zeros((n, n)), xbar=np.zeros(n), ubar=np.zeros(m),
                EPS=0.0, INF=0.0, ontarget_check=False)


class GridDynamicSystem:

    def __init__(self, sys, x_grid_dim=[101, 101], u_grid_dim=[11], dt=0.05, lookup=True):
        self.sys = sys
        self.dt = dt
        self.x_grid_dim = np.array(x_grid_dim)
        self.u_grid_dim = np.array(u_grid_dim)
        self.computelookuptable = lookup
        self.fontsize, self.figsize, self.dpi = 5, (4, 3), 300
        self._lazy = {}
        self.compute()

    # ------------------------------------------------------------------ set-up (discretizer.py:111-163)
    def compute(self):
        if self.sys.n not in (2, 3, 4) or self.sys.m not in (1, 2):
            raise NotImplementedError              # discretizer.py:245, :306
        if len(self.x_grid_dim) != self.sys.n or len(self.u_grid_dim) != self.sys.m:
            raise ValueError("grid dimensions do not match the system dimensions")
        self._lazy.clear()
        self.__dict__.pop("_trig", None)
        self.__dict__.pop("_trig_key", None)
        self.discretize_state_space()
        self.discretize_input_space()
        print("\nGenerating a mesh for:", self.sys.name)
        print("---------------------------------------------------")
        print("State space dimensions:", self.sys.n, " Input space dimension:", self.sys.m)
        print("Number of nodes:", self.nodes_n, " Number of actions:", self.actions_n)
        print("Number of node-action pairs:", self.nodes_n * self.actions_n)
        print("---------------------------------------------------")

    def discretize_state_space(self):
        s = self.sys
        self.x_level = [np.linspace(s.x_lb[i], s.x_ub[i], self.x_grid_dim[i]) for i in range(s.n)]
        self.nodes_n = int(np.prod(self.x_grid_dim))
        self.x_range = s.x_ub - s.x_lb
        self.x_step_size = self.x_range / (self.x_grid_dim - 1)

    def discretize_input_space(self):
        s = self.sys
        self.u_level = [np.linspace(s.u_lb[i], s.u_ub[i], self.u_grid_dim[i]) for i in range(s.m)]
        self.actions_n = int(np.prod(self.u_grid_dim))
        self.u_range = s.u_ub - s.u_lb
        self.u_step_size = self.u_range / (self.u_grid_dim - 1)

    # ------------------------------------------------------------------ enumeration (discretizer.py:167-310)
    @staticmethod
    def _enumerate(levels, dims):
        total = int(np.prod(dims))
        index = np.stack(np.unravel_index(np.arange(total), tuple(int(d) for d in dims)), axis=-1)
        value = np.stack([levels[k][index[:, k]] for k in range(len(dims))], axis=-1)
        ids = np.arange(total).reshape(tuple(int(d) for d in dims))
        return value, index.astype(int), ids

    def generate_nodes(self):
        v, i, ids = self._enumerate(self.x_level, self.x_grid_dim)
        self._lazy.update(state_from_node_id=v, index_from_node_id=i, node_id_from_index=ids)

    def generate_actions(self):
        v, i, ids = self._enumerate(self.u_level, self.u_grid_dim)
        self._lazy.update(input_from_action_id=v, index_from_action_id=i, action_id_from_index=ids)

    _NODE_ATTRS = ("state_from_node_id", "index_from_node_id", "node_id_from_index")
    _ACTION_ATTRS = ("input_from_action_id", "index_from_action_id", "action_id_from_index")
    _TABLE_ATTRS = ("x_next_table", "x_next_isok")

    def __getattr__(self, name):
        lazy = self.__dict__.get("_lazy")
        if lazy is None:
            raise AttributeError(name)
        if name not in lazy:
            if name in self._NODE_ATTRS:
                self.generate_nodes()
            elif name in self._ACTION_ATTRS:
                self.generate_actions()
            elif name in self._TABLE_ATTRS:
                self.compute_xnext_table()
            elif name == "action_isok":
                self.compute_action_set_table()
            else:
                raise AttributeError(name)
        return lazy[name]

    def __setattr__(self, name, value):
        if name in self._NODE_ATTRS + self._ACTION_ATTRS + self._TABLE_ATTRS + ("action_isok",):
            self._lazy[name] = value
        else:
            object.__setattr__(self, name, value)

    # ------------------------------------------------------------------ device handle for table builds
    def _device_problem(self, cost=None, dtype="float64", **kw):
        """libpyrovi problem for this grid; `cost` = dict from CostFunction.device_cost()."""
        return _native.Problem(**self._problem_kwargs(cost, dtype, **kw))

    def _shard_problem(self, rank, world, halo_rows, comm_id=None, overlap=True, cost=None, dtype="float32", transport=None,
                       **kw):
        """This rank's slab of the grid with the halo exchange inside the library (pvi_shard_*, RCCL)."""
        return _native.ShardedProblem(rank, world, halo_rows, comm_id=comm_id, overlap=overlap, transport=transport,
                                      **self._problem_kwargs(cost, dtype, **kw))

    def _problem_kwargs(self, cost=None, dtype="float64", **kw):
        s = self.sys
        dd = device_dynamics_of(s)
        if dd is None:
            dyn_id, params, trig = _native.DYN_TABLE, (), ()
        else:
            dyn_id, params = dd
            if dyn_id < _native.DYN_NODE_1x1:
                trig = s.device_trig(self.x_level)           # closed forms: a few np.sin / np.cos over the levels
            else:
                # per-node tables of generic mechanical systems are O(N) Python calls: cached, keyed on the grid
                # and on the system's parameters
                key = (dyn_id, _fingerprint(s, self.x_level, self.dt))
                if self.__dict__.get("_trig_key") != key:
                    t0 = time.time()
                    print("Computing per-node dynamics tables..  ", end="")
                    self.__dict__["_trig"] = s.device_trig(self.x_level)
                    self.__dict__["_trig_key"] = key
                    print("completed in %4.2f sec" % (time.time() - t0))
                trig = self.__dict__["_trig"]
        if dyn_id != _native.DYN_TABLE and cost is None:
            cost = _null_cost(s.n, s.m)
        if dd is not None and getattr(s.isavalidstate, "__func__", None) is not _base_isavalidstate():
            kw.setdefault("obstacles", s.device_obstacles())
        if dd is not None and hasattr(s, "device_act_aux"):
            kw.setdefault("act_aux", s.device_act_aux(self.input_from_action_id))
        return dict(x_levels=self.x_level, u_levels=self.u_level, x_lb=s.x_lb, x_ub=s.x_ub, u_lb=s.u_lb, u_ub=s.u_ub,
                    dt=self.dt, dtype=dtype, dynamics_id=dyn_id, dyn_params=params, trig=trig, cost=cost, **kw)

    # ------------------------------------------------------------------ look-up tables (discretizer.py:314-376)
    def compute_xnext_table(self):
        """x_next_table[s,a,:] = sys.f(x_s,u_a)*dt + x_s and x_next_isok[s,a] = sys.isavalidstate(x_next)."""
        t0 = time.time()
        print("Computing x_next array.. ", end="")
        if device_dynamics_of(self.sys) is not None:
            p = self._device_problem()
            xn, ok, _, _ = p.build_tables(action_isok=False, G=False)
            p.close()
        else:
            xn, ok = self._host_xnext_table()
        self._lazy.update(x_next_table=xn, x_next_isok=ok)
        print("completed in %4.2f sec" % (time.time() - t0))

    def _xnext_rows(self, lo, hi):
        """The reference's own loop over sys.f (arbitrary Python) for nodes [lo, hi)."""
        s, X, U = self.sys, self.state_from_node_id, self.input_from_action_id
        xn = np.zeros((hi - lo, self.actions_n, s.n))
        ok = np.zeros((hi - lo, self.actions_n), dtype=bool)
        for i in range(lo, hi):
            for a in range(self.actions_n):
                x_next = s.f(X[i], U[a]) * self.dt + X[i]
                xn[i - lo, a] = x_next
                ok[i - lo, a] = s.isavalidstate(x_next)
        return xn, ok

    def _host_xnext_table(self):
        """Generic systems: x_next / isok by calling sys.f and sys.isavalidstate cell by cell, as the reference does
        (discretizer.py:342-376) -- the calls are independent, so large tables are split over the host cores."""
        parts = host_parallel_rows(self, "_xnext_rows", self.nodes_n, self.nodes_n * self.actions_n)
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    def compute_action_set_table(self):
        """action_isok[s,a] = sys.isavalidinput(x_s, u_a)."""
        s = self.sys
        if type(s).isavalidinput is _box_isavalidinput():
            U = self.input_from_action_id
            row = ~(np.any(U < s.u_lb, axis=1) | np.any(U > s.u_ub, axis=1))
            ok = np.broadcast_to(row, (self.nodes_n, self.actions_n)).copy()
        else:
            X, U = self.state_from_node_id, self.input_from_action_id
            ok = np.array([[s.isavalidinput(X[i], U[a]) for a in range(self.actions_n)]
                           for i in range(self.nodes_n)], dtype=bool)
        self._lazy["action_isok"] = ok

    def save_lookup_tables(self, name="grid"):
        np.savez(name, x_next_table=self.x_next_table, x_next_isok=self.x_next_isok, action_isok=self.action_isok)

    def load_lookup_tables(self, name="grid"):
        try:
            data = np.load(name + ".npz")
        except Exception:
            print("\n File not found ")
        else:
            self._lazy.update(x_next_table=data["x_next_table"], x_next_isok=data["x_next_isok"],
                              action_isok=data["action_isok"])

    # ------------------------------------------------------------------ index helpers (discretizer.py:453-537)
    def get_index_from_state(self, x):
        return (np.asarray(x, dtype=float) - self.sys.x_lb) / self.x_range * (self.x_grid_dim - 1)

    def get_nearest_index_from_state(self, x):
        return np.clip(np.rint(self.get_index_from_state(x)).astype(int), 0, self.x_grid_dim - 1)

    def get_nearest_node_id_from_state(self, x):
        return int(np.ravel_multi_index(tuple(self.get_nearest_index_from_state(x)), tuple(self.x_grid_dim)))

    def get_index_from_input(self, u):
        return (np.asarray(u, dtype=float) - self.sys.u_lb) / self.u_range * (self.u_grid_dim - 1)

    def get_nearest_index_from_input(self, u):
        return np.clip(np.rint(self.get_index_from_input(u)).astype(int), 0, self.u_grid_dim - 1)

    def get_nearest_action_id_from_input(self, u):
        return int(np.ravel_multi_index(tuple(self.get_nearest_index_from_input(u)), tuple(self.u_grid_dim)))

    # ------------------------------------------------------------------ tools (discretizer.py:545-633)
    def get_grid_from_array(self, J):
        return J.reshape(self.x_grid_dim)

    def compute_interpolation_function(self, J, method="linear", bounds_error=True, fill_value=None):
        if self.nodes_n != J.size:
            raise ValueError("Grid size does not match data")
        return RegularGridInterpolator(tuple(self.x_level), self.get_grid_from_array(J), method, bounds_error,
                                       fill_value)

    def compute_bivariatespline_2D_interpolation_function(self, J, kx=1, ky=1):
        if self.sys.n != 2:
            raise NotImplementedError
        if self.nodes_n != J.size:
            raise ValueError("Grid size does not match data")
        return RectBivariateSpline(self.x_level[0], self.x_level[1], self.get_grid_from_array(J),
                                   bbox=[None, None, None, None], kx=kx, ky=ky)

    def get_input_from_policy(self, pi, k):
        if self.nodes_n != pi.size:
            raise ValueError("Grid size does not match optimal action table size")
        return self.input_from_action_id[np.asarray(pi, dtype=np.int64), k].astype(float)

    def get_2D_slice_of_grid(self, Z, axis_1=0, axis_2=1):
        if self.sys.n == 2:
            return Z
        idx = [int(i) for i in self.get_nearest_index_from_state(self.sys.xbar)]
        idx[axis_1], idx[axis_2] = slice(None), slice(None)
        out = np.asarray(Z[tuple(idx)], dtype=float)
        return out if axis_1 < axis_2 else out.T

    # ------------------------------------------------------------------ plots: host pass-through (matplotlib)
    def _slice_2d(self, J, x, y, jmin, jmax):
        Z = self.get_2D_slice_of_grid(self.get_grid_from_array(np.asarray(J, dtype=float)), x, y)
        return np.clip(Z, jmin, jmax)

    def _label_axes(self, ax, x, y):
        s = self.sys
        ax.set_xlabel(s.state_label[x] + " " + s.state_units[x], fontsize=self.fontsize)
        ax.set_ylabel(s.state_label[y] + " " + s.state_units[y], fontsize=self.fontsize)
        ax.tick_params(labelsize=self.fontsize)

    def plot_grid_value(self, J, name="Value on the grid", x=0, y=1, jmax=np.inf, jmin=-1, cmap="YlOrRd"):
        """2-D colour map of a node array over state axes (x, y); other axes sliced at sys.xbar
        (discretizer.py:668-735).  Returns (fig, ax, mesh)."""
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots(figsize=self.figsize, dpi=self.dpi, frameon=True)
        fig.canvas.manager.set_window_title(name)
        mesh = ax.pcolormesh(self.x_level[x], self.x_level[y], self._slice_2d(J, x, y, jmin, jmax).T,
                             shading="gouraud", cmap=cmap)
        self._label_axes(ax, x, y)
        ax.grid(True)
        cbar = fig.colorbar(mesh, ax=ax)
        cbar.ax.tick_params(labelsize=self.fontsize)
        fig.tight_layout()
        return fig, ax, mesh

    def plot_grid_value_3D(self, J, J2=None, name="Value on the grid", x=0, y=1, jmax=np.inf, jmin=-1, cmap="YlOrRd"):
        """Surface plot of one (or two) node arrays (discretizer.py:739-823).  Returns (fig, ax, surf)."""
        import matplotlib.pyplot as plt
        fig = plt.figure(figsize=self.figsize, dpi=self.dpi)
        ax = fig.add_subplot(projection="3d")
        fig.canvas.manager.set_window_title(name)
        X, Y = np.meshgrid(self.x_level[x], self.x_level[y], indexing="ij")
        surf = ax.plot_surface(X, Y, self._slice_2d(J, x, y, jmin, jmax), cmap=cmap, linewidth=0, antialiased=False)
        if J2 is not None:
            ax.plot_surface(X, Y, self._slice_2d(J2, x, y, jmin, jmax), alpha=0.5, linewidth=0)
        self._label_axes(ax, x, y)
        ax.set_zlabel(name, fontsize=self.fontsize)
        fig.tight_layout()
        return fig, ax, surf

    def plot_control_input_from_policy(self, pi, k, i=0, j=1):
        """Colour map of input axis k selected by the policy (discretizer.py:826-834)."""
        return self.plot_grid_value(self.get_input_from_policy(pi, k), self.sys.input_label[k], i, j,
                                    self.sys.u_ub[k], self.sys.u_lb[k], cmap="bwr")


# ---- host-side parallelism for the O(N*A) Python loops of the table tier -------------------------------------------
_FORK_TARGET = {}


def _fork_call(args):
    name, lo, hi = args
    return getattr(_FORK_TARGET["obj"], name)(lo, hi)


def host_parallel_rows(obj, method, n_rows, n_calls, min_calls=200000):
    """[obj.method(lo, hi) for consecutive row blocks], in forked worker processes when the job is large.
    The workers inherit `obj` (system, cost function: arbitrary Python, not necessarily picklable) through fork and
    never touch the GPU.  PYRO_AMD_HOST_WORKERS sets the worker count (0 or 1: serial)."""
    import multiprocessing as mp
    import os
    workers = int(os.environ.get("PYRO_AMD_HOST_WORKERS", min(os.cpu_count() or 1, 64)))
    if workers <= 1 or n_calls < min_calls or n_rows < 2 * workers:
        return [getattr(obj, method)(0, n_rows)]
    bounds = np.linspace(0, n_rows, 4 * workers + 1).astype(int)
    jobs = [(method, int(a), int(b)) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    _FORK_TARGET["obj"] = obj
    try:
        with mp.get_context("fork").Pool(workers) as pool:
            return pool.map(_fork_call, jobs)
    except Exception:                                   # no fork / no semaphores: the plain loop always works
        return [getattr(obj, method)(0, n_rows)]
    finally:
        _FORK_TARGET.pop("obj", None)


def device_cost_of(cost_function, sys):
    """The in-kernel description of `cost_function` on `sys` (dict for _native.Problem), or None when the cost has to go
    through the look-up tables.  A domain-check / reachability cost tests states against a SYSTEM's validity: in-kernel
    that is the validity of the grid's own system (box + its obstacle list), so the cost must be bound to exactly that
    system's stock isavalidstate -- one bound to another system, or to a wrapped test, is arbitrary Python."""
    cost = cost_function.device_cost() if hasattr(cost_function, "device_cost") else None
    if cost is not None and cost.get("kind") in ("quadratic_domain", "reachability"):
        test = cost_function.isavalidstate if cost["kind"] == "quadratic_domain" else cost_function.isavalidestate
        if cost.pop("validity_of", None) is not sys or getattr(test, "__func__", None) is not getattr(sys.isavalidstate, "__func__", 0):
            return None
    return cost


def device_dynamics_of(sys):
    """(dynamics_id, params) when `sys` can be evaluated in-kernel: it must say so itself
    (device_dynamics()) AND keep the plain box validity tests the kernels implement."""
    from pyro_amd.dynamic.system import ContinuousDynamicSystem as Base
    fn = getattr(sys, "device_dynamics", None)
    if fn is None or not isinstance(sys, Base):
        return None
    # the kernels implement the plain inclusive box: a subclass override AND an instance attribute (the reference
    # itself assigns sys.isavalidstate on instances, manipulator.py:441) both send the system to the table tier
    vi = getattr(sys.isavalidinput, "__func__", None)
    if vi is not Base.isavalidinput:
        # a state-dependent input test is in-kernel only as the stock test of the class the kernel was written for
        owner = getattr(type(sys), "_INPUT_VALIDITY_OWNER", None)
        if owner is None or vi is not owner.__dict__.get("isavalidinput"):
            return None
    vs = getattr(sys.isavalidstate, "__func__", None)
    if vs is not Base.isavalidstate:
        # box + obstacle boxes: in-kernel when it is exactly the test of the class that describes its obstacles to the
        # library (device_obstacles of the helicopter tunnel / the car with obstacles)
        owner = getattr(type(sys), "_OBSTACLE_OWNER", None)
        if owner is None or vs is not owner.__dict__.get("isavalidstate") or "device_obstacles" in vars(sys):
            return None
        if type(sys).device_obstacles is not owner.device_obstacles:
            return None
        if len(np.asarray(sys.device_obstacles()["boxes"], dtype=float).reshape(-1, 4)) > _native.PVI_MAX_OBS:
            return None             # more boxes than the descriptor holds: the look-up tables take any number
    return fn()


def _fingerprint(sys, x_level, dt):
    """Key of the cached per-node dynamics tables: grid levels + every numeric attribute of the system (mass, bounds,
    ...), so that editing the system or the grid between two DynamicProgramming objects rebuilds them."""
    import hashlib
    h = hashlib.sha1()
    for l in x_level:
        h.update(np.ascontiguousarray(l, dtype=np.float64).tobytes())
    h.update(repr((type(sys).__qualname__, float(dt))).encode())
    for k in sorted(vars(sys)):
        v = vars(sys)[k]
        if isinstance(v, (bool, int, float, np.integer, np.floating)):
            h.update(("%s=%r;" % (k, float(v))).encode())
        elif isinstance(v, np.ndarray) and v.dtype.kind in "fiub" and v.size <= 4096:
            h.update(k.encode() + np.ascontiguousarray(v, dtype=np.float64).tobytes())
    return h.hexdigest()


def _base_isavalidstate():
    from pyro_amd.dynamic.system import ContinuousDynamicSystem
    return ContinuousDynamicSystem.isavalidstate


def _box_isavalidinput():
    from pyro_amd.dynamic.system import ContinuousDynamicSystem
    return ContinuousDynamicSystem.isavalidinput
