"""
ctypes binding of libpyrovi.so (C ABI: include/pyrovi.h).

There is NO CPU fallback: if the shared library is missing, or a call fails, a RuntimeError is
raised.  The library itself loads on a machine without a GPU (so symbol checks work there);
creating a problem needs a HIP device.
"""
import ctypes as C
import os

import numpy as np

PVI_MAX_N, PVI_MAX_M, PVI_MAX_TRIG = 4, 2, 4
PVI_F32, PVI_F64 = 0, 1
DYN_TABLE, DYN_PENDULUM, DYN_CARTPOLE, DYN_TWOLINK = 0, 1, 2, 3
DYN_NODE_1x1, DYN_NODE_2x1, DYN_NODE_2x2 = 4, 5, 6        # any MechanicalSystem through per-node tables
DYN_HELICOPTER, DYN_KINCAR, DYN_QUARTERCAR = 7, 8, 9      # the reference's three-dimensional demo systems (n = 3)
DYN_HOLONOMIC, DYN_LONGCAR = 10, 11                       # point robot with obstacles, longitudinal car (n = 2)
DYN_CARTPOLE_SW = 12                                      # opt-in: the cart-pole with q = (theta, x) (float32 4-D window sweep only)
CLOSED_FORM_IDS = (DYN_PENDULUM, DYN_CARTPOLE, DYN_TWOLINK)    # dynamics pvi_eval_f can evaluate anywhere
# ... and pvi_rollout: the explicit systems have continuous closed forms too (their sweeps read host tables at the nodes)
ROLLOUT_IDS = CLOSED_FORM_IDS + (DYN_HELICOPTER, DYN_KINCAR, DYN_QUARTERCAR, DYN_HOLONOMIC, DYN_LONGCAR)
COST_TABLE, COST_QUADRATIC, COST_TIME, COST_QUADRATIC_DOMAIN, COST_REACHABILITY = 0, 1, 2, 3, 4
INTERP_LINEAR, INTERP_BICUBIC_SPLINE, INTERP_NEAREST = 0, 1, 2
CTL_TABLE, CTL_COMPUTED_TORQUE = 0, 1
PVI_EHALO = -5
PVI_ECORRUPT = -6      # the corruption detector of the error-feedback sweep fired (pvi_override FBCHECK=1)
FLAG_EXT_J_SLACK = 1
FLAG_HARD_INF = 2
FLAG_F32_FEEDBACK = 4     # error-feedback storage of a float32 J (4-D window sweep; include/pyrovi.h)
ABI_VERSION = 3
PVI_MAX_OBS = 8

_dp = C.POINTER(C.c_double)


class pvi_desc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n", C.c_int32), ("m", C.c_int32),
        ("x_dim", C.c_int32 * PVI_MAX_N), ("u_dim", C.c_int32 * PVI_MAX_M),
        ("x_level", _dp * PVI_MAX_N), ("u_level", _dp * PVI_MAX_M),
        ("x_lb", C.c_double * PVI_MAX_N), ("x_ub", C.c_double * PVI_MAX_N),
        ("u_lb", C.c_double * PVI_MAX_M), ("u_ub", C.c_double * PVI_MAX_M),
        ("dt", C.c_double), ("dtype", C.c_int32), ("dynamics_id", C.c_int32),
        ("dyn_params", C.c_double * 16), ("trig", _dp * PVI_MAX_TRIG),
        ("cost_id", C.c_int32), ("ontarget_check", C.c_int32),
        ("Q", C.c_double * 16), ("R", C.c_double * 4), ("S", C.c_double * 16),
        ("xbar", C.c_double * PVI_MAX_N), ("ubar", C.c_double * PVI_MAX_M),
        ("EPS", C.c_double), ("INF", C.c_double),
        ("row_begin", C.c_int32), ("row_end", C.c_int32), ("halo_lo", C.c_int32), ("halo_hi", C.c_int32),
        ("device", C.c_int32), ("flags", C.c_int32),
        ("ext_J", C.c_void_p * 2), ("ext_pi", C.c_void_p),
        # ABI 2: obstacle boxes of isavalidstate, per-action constants of the dynamics
        ("n_obs", C.c_int32), ("obs_axis", C.c_int32 * 2), ("obs_half", C.c_double * 2),
        ("obs_box", (C.c_double * 4) * PVI_MAX_OBS), ("act_aux", _dp),
    ]


# every symbol include/pyrovi.h declares: name -> (restype, argtypes)
_h = C.c_void_p
SYMBOLS = {
    "pvi_abi_version": (C.c_int, []),
    "pvi_last_error": (C.c_char_p, []),
    "pvi_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pvi_override": (C.c_int, [C.c_char_p, C.c_char_p]),
    "pvi_create": (C.c_int, [C.POINTER(pvi_desc), C.POINTER(_h)]),
    "pvi_destroy": (None, [_h]),
    "pvi_plane_size": (C.c_int64, [_h]),
    "pvi_stored_nodes": (C.c_int64, [_h]),
    "pvi_owned_nodes": (C.c_int64, [_h]),
    "pvi_pi_itemsize": (C.c_int, [_h]),
    "pvi_describe": (C.c_int, [_h, C.c_char_p, C.c_int32]),
    "pvi_terminal_cost": (C.c_int, [_h]),
    "pvi_set_J": (C.c_int, [_h, _dp, C.c_int32, C.c_int32]),
    "pvi_get_J": (C.c_int, [_h, _dp, C.c_int32, C.c_int32]),
    "pvi_get_J_prev": (C.c_int, [_h, _dp, C.c_int32, C.c_int32]),
    "pvi_get_pi": (C.c_int, [_h, C.POINTER(C.c_int64), C.c_int32, C.c_int32]),
    "pvi_sweep": (C.c_int, [_h, C.c_int32, C.c_double, C.c_double, _dp, C.POINTER(C.c_int32)]),
    "pvi_last_sweep_ms": (C.c_int, [_h, C.POINTER(C.c_float)]),
    "pvi_sweep_async": (C.c_int, [_h, C.c_double, C.c_void_p]),
    "pvi_sweep_stats": (C.c_int, [_h, _dp, C.c_void_p]),
    "pvi_device_J": (C.c_int, [_h, C.c_int, C.POINTER(C.c_void_p)]),
    "pvi_device_pi": (C.c_int, [_h, C.POINTER(C.c_void_p)]),
    "pvi_synchronize": (C.c_int, [_h]),
    "pvi_self_check": (C.c_int, [_h, C.c_double, _dp, C.POINTER(C.c_int64)]),
    "pvi_plan_plane_tiles": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32,
                                       C.POINTER(C.c_int32), C.c_int32]),
    "pvi_plan_schedule": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.c_int64]),
    "pvi_plan_schedule_rows": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_int64]),
    "pvi_build_tables": (C.c_int, [_h, C.c_int32, C.c_int32, _dp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), _dp]),
    "pvi_set_tables": (C.c_int, [_h, _dp, _dp, C.POINTER(C.c_uint8)]),
    "pvi_policy_tables": (C.c_int, [_h, C.c_int32, _dp, _dp, _dp, C.POINTER(C.c_uint8), _dp]),
    "pvi_set_interpolation": (C.c_int, [_h, C.c_int32]),
    "pvi_spline_coefficients": (C.c_int, [_h, _dp]),
    "pvi_set_pi": (C.c_int, [_h, C.POINTER(C.c_int64), C.c_int32, C.c_int32]),
    "pvi_rollout": (C.c_int, [_h, C.c_int64, _dp, C.c_int32, C.c_double, _dp, _dp, _dp]),
    "pvi_set_rollout_params": (C.c_int, [_h, _dp, C.c_int32]),
    "pvi_eval_f": (C.c_int, [C.c_int32, _dp, C.c_int32, C.c_int32, C.c_int64, _dp, _dp, _dp]),
    # multi-GPU: slabs of axis 0 with RCCL inside the library
    "pvi_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "pvi_shard_create": (C.c_int, [C.POINTER(pvi_desc), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.c_int32,
                                   C.POINTER(_h)]),
    "pvi_shard_create_with_transport": (C.c_int, [C.POINTER(pvi_desc), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                                  C.POINTER(_h)]),
    "pvi_shard_destroy": (None, [_h]),
    "pvi_shard_rows": (C.c_int, [_h, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pvi_shard_terminal_cost": (C.c_int, [_h]),
    "pvi_shard_sweep": (C.c_int, [_h, C.c_int32, C.c_double, C.c_double, _dp, C.POINTER(C.c_int32)]),
    "pvi_shard_set_tables": (C.c_int, [_h, _dp, _dp, C.POINTER(C.c_uint8)]),
    "pvi_shard_set_J": (C.c_int, [_h, _dp]),
    "pvi_shard_get_J": (C.c_int, [_h, _dp]),
    "pvi_shard_get_J_prev": (C.c_int, [_h, _dp]),
    "pvi_shard_halo": (C.c_int, [_h, C.POINTER(C.c_int32)]),
    "pvi_shard_get_pi": (C.c_int, [_h, C.POINTER(C.c_int64)]),
    "pvi_shard_describe": (C.c_int, [_h, C.c_char_p, C.c_int32]),
    "pvi_shard_timing": (C.c_int, [_h, _dp]),
    "pvi_shard_gather_J": (C.c_int, [_h, C.c_int32, _dp]),
    "pvi_shard_gather_pi": (C.c_int, [_h, C.POINTER(C.c_int64)]),
    "pvi_shard_stats_every_sweep": (C.c_int, [_h, C.c_int32]),
    "pvi_shard_sweep_history": (C.c_int, [_h, _dp, C.c_int32, C.POINTER(C.c_int32)]),
}
COMM_ID_BYTES = 128

# (PYROVI_LIB: another build of the same library, e.g. the sanitizer host build pyro_amd/libpyrovi_ubsan.so)
LIB_PATH = os.environ.get("PYROVI_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpyrovi.so")
_lib = None


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libpyrovi error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load libpyrovi.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "pyro_amd: %s is missing -- build it with `python -m pyro_amd._build` "
                "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)            # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if L.pvi_abi_version() != ABI_VERSION:
            raise RuntimeError("libpyrovi ABI %d != binding ABI %d" % (L.pvi_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise NativeError(rc, lib().pvi_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    check(lib().pvi_device_count(C.byref(n)))
    return n.value


def override(key=None, value=None):
    """Pin a kernel variant for handles created from now on (pvi_override; tests and profiling).  override(key, None)
    removes the key, override() removes all."""
    check(lib().pvi_override(None if key is None else str(key).encode(), None if value is None else str(value).encode()))
    if key is None:
        _OVERRIDES.clear()
    elif value is None:
        _OVERRIDES.pop(str(key), None)
    else:
        _OVERRIDES[str(key)] = str(value)


_OVERRIDES = {}          # what this process has pinned through override() (the library keeps its own copy)


def override_value(key):
    """The value override(key, value) pinned in this process, or None."""
    return _OVERRIDES.get(str(key))


class overrides:
    """with _native.overrides(LSPLIT=0, NPT=2): ...   -- the keys are cleared again on exit."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        for k, v in self.kv.items():
            override(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kv:
            override(k, None)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def eval_f(dyn_id, params16, X, U):
    """Batched dx = f(x,u) on the GPU (pvi_eval_f)."""
    X, U = _f64(np.atleast_2d(X)), _f64(np.atleast_2d(U))
    p = np.zeros(16)
    p[:len(params16)] = params16
    out = np.empty_like(X)
    check(lib().pvi_eval_f(dyn_id, _ptr(p), X.shape[1], U.shape[1], X.shape[0], _ptr(X), _ptr(U), _ptr(out)))
    return out


class Problem:
    """Owner of one pvi_handle.  All arrays are host NumPy; see include/pyrovi.h."""

    def __init__(self, x_levels, u_levels, x_lb, x_ub, u_lb, u_ub, dt, dtype="float64", dynamics_id=DYN_TABLE,
                 dyn_params=(), trig=(), cost=None, rows=None, halo=(0, 0), device=0, ext_J=None, ext_pi=None,
                 table_inf=0.0, flags=0, obstacles=None, act_aux=None, _create=True):
        L = lib()
        self._keep = []                      # host buffers the descriptor points to
        d = pvi_desc()
        d.struct_size = C.sizeof(pvi_desc)
        d.n, d.m = len(x_levels), len(u_levels)
        self.n, self.m = d.n, d.m
        self.dims = tuple(len(l) for l in x_levels)
        self.u_dims = tuple(len(l) for l in u_levels)
        for i, l in enumerate(x_levels):
            a = _f64(l); self._keep.append(a)
            d.x_dim[i], d.x_level[i] = len(a), _ptr(a)
            d.x_lb[i], d.x_ub[i] = float(x_lb[i]), float(x_ub[i])
        for k, l in enumerate(u_levels):
            a = _f64(l); self._keep.append(a)
            d.u_dim[k], d.u_level[k] = len(a), _ptr(a)
            d.u_lb[k], d.u_ub[k] = float(u_lb[k]), float(u_ub[k])
        d.dt = float(dt)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype("float32"), np.dtype("float64")):
            raise ValueError("dtype must be float32 or float64")
        d.dtype = PVI_F64 if self.dtype == np.dtype("float64") else PVI_F32
        d.dynamics_id = int(dynamics_id)
        self.dynamics_id = int(dynamics_id)
        for i, v in enumerate(dyn_params):
            d.dyn_params[i] = float(v)
        for i, t in enumerate(trig):
            if t is not None:
                a = _f64(t); self._keep.append(a)
                d.trig[i] = _ptr(a)
        if cost is not None and cost.get("kind") in ("time", "reachability"):
            d.cost_id = COST_TIME if cost["kind"] == "time" else COST_REACHABILITY
            d.xbar[:d.n] = [float(v) for v in cost["xbar"]]
            d.EPS, d.INF = float(cost["EPS"]), float(cost["INF"])
            d.ontarget_check = int(bool(cost.get("ontarget_check", True)))
        elif cost is not None:
            d.cost_id = COST_QUADRATIC_DOMAIN if cost.get("kind") == "quadratic_domain" else COST_QUADRATIC
            n, m = d.n, d.m
            for name, k in (("Q", n), ("S", n), ("R", m)):
                M = _f64(cost[name])
                if M.shape != (k, k):
                    raise ValueError("cost %s must be %dx%d" % (name, k, k))
                getattr(d, name)[:k * k] = list(M.ravel())
            d.xbar[:n] = [float(v) for v in cost["xbar"]]
            d.ubar[:m] = [float(v) for v in cost["ubar"]]
            d.EPS, d.INF = float(cost["EPS"]), float(cost["INF"])
            d.ontarget_check = int(bool(cost.get("ontarget_check", True)))
        else:
            d.cost_id = COST_TABLE
            d.INF = float(table_inf)
        r0, r1 = (0, self.dims[0]) if rows is None else rows
        d.row_begin, d.row_end, d.halo_lo, d.halo_hi = int(r0), int(r1), int(halo[0]), int(halo[1])
        d.device = int(device)
        if ext_J is not None:
            d.ext_J[0], d.ext_J[1] = int(ext_J[0]), int(ext_J[1])
        if ext_pi is not None:
            d.ext_pi = int(ext_pi)
        d.flags = int(flags)
        if obstacles is not None:
            # dict(axes=(ax, ay), half=(hx, hy), boxes=[[lo_x, lo_y, hi_x, hi_y], ...]) -- include/pyrovi.h obs_*
            boxes = np.asarray(obstacles["boxes"], dtype=np.float64).reshape(-1, 4)
            if len(boxes) > PVI_MAX_OBS:
                raise ValueError("at most %d obstacle boxes" % PVI_MAX_OBS)
            d.n_obs = len(boxes)
            d.obs_axis[0], d.obs_axis[1] = int(obstacles["axes"][0]), int(obstacles["axes"][1])
            d.obs_half[0], d.obs_half[1] = float(obstacles["half"][0]), float(obstacles["half"][1])
            for b, box in enumerate(boxes):
                for k in range(4):
                    d.obs_box[b][k] = float(box[k])
        if act_aux is not None:
            a = _f64(act_aux); self._keep.append(a)
            d.act_aux = _ptr(a)
        self.rows = (int(r0), int(r1))
        self.store_rows = (max(0, r0 - halo[0]), min(self.dims[0], r1 + halo[1]))
        self._h = _h()
        self._desc = d
        self.actions_n = int(np.prod(self.u_dims))
        if not _create:                      # descriptor only (ShardedProblem hands it to pvi_shard_create)
            self.plane = int(np.prod(self.dims[1:]))
            return
        check(L.pvi_create(C.byref(d), C.byref(self._h)))
        self.plane = L.pvi_plane_size(self._h)
        self.owned_nodes = L.pvi_owned_nodes(self._h)
        self.stored_nodes = L.pvi_stored_nodes(self._h)
        self.actions_n = int(np.prod(self.u_dims))

    # ------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib().pvi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _rows(self, row0, nrows, default):
        if row0 is None:
            row0, nrows = default[0], default[1] - default[0]
        return int(row0), int(nrows)

    def describe(self):
        buf = C.create_string_buffer(4096)
        check(lib().pvi_describe(self._h, buf, 4096))
        return buf.value.decode()

    def terminal_cost(self):
        check(lib().pvi_terminal_cost(self._h))

    def set_J(self, J, row0=None, nrows=None):
        row0, nrows = self._rows(row0, nrows, self.store_rows)
        J = _f64(J).ravel()
        if J.size != nrows * self.plane:
            raise ValueError("Grid size does not match data")
        check(lib().pvi_set_J(self._h, _ptr(J), row0, nrows))

    def get_J(self, row0=None, nrows=None, prev=False):
        row0, nrows = self._rows(row0, nrows, self.rows)
        out = np.empty(nrows * self.plane, dtype=np.float64)
        fn = lib().pvi_get_J_prev if prev else lib().pvi_get_J
        check(fn(self._h, _ptr(out), row0, nrows))
        return out

    def get_pi(self, row0=None, nrows=None):
        row0, nrows = self._rows(row0, nrows, self.rows)
        out = np.empty(nrows * self.plane, dtype=np.int64)
        check(lib().pvi_get_pi(self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), row0, nrows))
        return out

    def set_pi(self, pi, row0=None, nrows=None):
        row0, nrows = self._rows(row0, nrows, self.rows)
        pi = np.ascontiguousarray(pi, dtype=np.int64).ravel()
        if pi.size != nrows * self.plane:
            raise ValueError("Grid size does not match optimal action table size")
        check(lib().pvi_set_pi(self._h, pi.ctypes.data_as(C.POINTER(C.c_int64)), row0, nrows))

    def set_rollout_params(self, params):
        """Constants of the system's CONTINUOUS closed form for pvi_rollout (include/pyrovi.h pvi_set_rollout_params)."""
        p = _f64(params).ravel()
        check(lib().pvi_set_rollout_params(self._h, _ptr(p), int(p.size)))

    def rollout(self, X0, npts, dt, trajectory=True):
        """Closed-loop Euler rollouts of the device policy.  Returns (X [B,npts,n], U [B,npts,m]) or X_end [B,n]."""
        X0 = _f64(np.atleast_2d(X0))
        B = X0.shape[0]
        if trajectory:
            X = np.empty((B, npts, self.n)); U = np.empty((B, npts, self.m))
            check(lib().pvi_rollout(self._h, B, _ptr(X0), int(npts), float(dt), _ptr(X), _ptr(U), None))
            return X, U
        Xe = np.empty((B, self.n))
        check(lib().pvi_rollout(self._h, B, _ptr(X0), int(npts), float(dt), None, None, _ptr(Xe)))
        return Xe

    def sweep(self, max_sweeps, alpha=1.0, tol=-1.0):
        """Returns (stats[k,4] = max J, dmax, dmin, delta ; sweeps_done)."""
        stats = np.zeros((max(int(max_sweeps), 1), 4), dtype=np.float64)
        done = C.c_int32(0)
        check(lib().pvi_sweep(self._h, int(max_sweeps), float(alpha), float(tol), _ptr(stats), C.byref(done)))
        return stats[:done.value], done.value

    def last_sweep_ms(self):
        ms = C.c_float(0)
        check(lib().pvi_last_sweep_ms(self._h, C.byref(ms)))
        return ms.value

    def sweep_async(self, alpha=1.0, stream=None):
        check(lib().pvi_sweep_async(self._h, float(alpha), C.c_void_p(stream or 0)))

    def sweep_stats(self, stream=None):
        out = np.zeros(3)
        check(lib().pvi_sweep_stats(self._h, _ptr(out), C.c_void_p(stream or 0)))
        return out

    def device_J(self, which=0):
        p = C.c_void_p()
        check(lib().pvi_device_J(self._h, int(which), C.byref(p)))
        return p.value

    def device_pi(self):
        p = C.c_void_p()
        check(lib().pvi_device_pi(self._h, C.byref(p)))
        return p.value

    def synchronize(self):
        check(lib().pvi_synchronize(self._h))

    def self_check(self, alpha=1.0):
        """(max relative difference of J, nodes whose action differs) between the production kernel path and the
        plain-gather kernel, for one backup of the current cost-to-go (pvi_self_check)."""
        d, n = C.c_double(0), C.c_int64(0)
        check(lib().pvi_self_check(self._h, float(alpha), C.byref(d), C.byref(n)))
        return d.value, n.value

    def build_tables(self, row0=None, nrows=None, x_next=True, x_next_isok=True, action_isok=True, G=True):
        row0, nrows = self._rows(row0, nrows, (0, self.dims[0]))
        nodes, A = nrows * self.plane, self.actions_n
        xn = np.empty((nodes, A, self.n)) if x_next else None
        xo = np.empty((nodes, A), dtype=np.uint8) if x_next_isok else None
        ao = np.empty((nodes, A), dtype=np.uint8) if action_isok else None
        g = np.empty((nodes, A)) if G else None
        u8 = C.POINTER(C.c_uint8)
        check(lib().pvi_build_tables(
            self._h, row0, nrows, _ptr(xn) if x_next else None,
            xo.ctypes.data_as(u8) if x_next_isok else None, ao.ctypes.data_as(u8) if action_isok else None,
            _ptr(g) if G else None))
        return xn, (xo.astype(bool) if xo is not None else None), (ao.astype(bool) if ao is not None else None), g

    def policy_tables(self, controller_id, ctl_params=None, U=None):
        """(U [nodes, m], x_next [nodes, n], ok [nodes], G [nodes]) of one control input per node (pvi_policy_tables)."""
        nodes = self.dims[0] * self.plane
        if controller_id == CTL_TABLE:
            U = np.ascontiguousarray(np.asarray(U, dtype=np.float64).reshape(nodes, self.m))
        else:
            U = np.empty((nodes, self.m))
        cp = None if ctl_params is None else _f64(ctl_params)
        xn, G = np.empty((nodes, self.n)), np.empty(nodes)
        ok = np.empty(nodes, dtype=np.uint8)
        check(lib().pvi_policy_tables(self._h, int(controller_id), None if cp is None else _ptr(cp), _ptr(U), _ptr(xn),
                                      ok.ctypes.data_as(C.POINTER(C.c_uint8)), _ptr(G)))
        return U, xn, ok.astype(bool), G

    def set_tables(self, x_next, G, ok=None):
        """ok=None: look-up-table semantics (INF + alpha*J); ok mask: base-class semantics (exactly INF)."""
        x_next, G = _f64(x_next), _f64(G)
        A = self.actions_n
        if x_next.shape != (self.owned_nodes, A, self.n) or G.shape != (self.owned_nodes, A):
            raise ValueError("table shapes do not match the grid")
        okp = None
        if ok is not None:
            ok = np.ascontiguousarray(ok, dtype=np.uint8)
            if ok.shape != G.shape:
                raise ValueError("ok mask shape does not match the grid")
            okp = ok.ctypes.data_as(C.POINTER(C.c_uint8))
        check(lib().pvi_set_tables(self._h, _ptr(x_next), _ptr(G), okp))

    def set_interpolation(self, kind):
        """'linear' (RegularGridInterpolator, default), 'nearest' (the same with method='nearest'; table tier, before the
        tables are set) or 'bicubic' (RectBivariateSpline kx=ky=3, 2-D grids)."""
        code = {"linear": INTERP_LINEAR, "bicubic": INTERP_BICUBIC_SPLINE, "nearest": INTERP_NEAREST}.get(kind, kind)
        check(lib().pvi_set_interpolation(self._h, int(code)))

    def spline_coefficients(self):
        """B-spline coefficients of the spline through the current J, shape x_dims (get_coeffs() order)."""
        out = np.empty(tuple(self.dims), dtype=np.float64)
        check(lib().pvi_spline_coefficients(self._h, _ptr(out)))
        return out


SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                          C.c_size_t, C.c_size_t, C.c_void_p)
MAX3_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))


class pvi_transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("sendrecv", SENDRECV_FN), ("max3", MAX3_FN)]


def plan_plane_tiles(V0, V1, corner0, cap, threads, wmax=None):
    """Host-only diagnostic (pvi_plan_plane_tiles): the rectangles {row0, nrows, col0, ncols} the 4-D float32 sweep would cut
    one V0 x V1 velocity plane into; corner0[j] = axis-0 corner index reached from velocity row j."""
    c0 = np.ascontiguousarray(corner0, dtype=np.int32)
    if c0.shape != (int(V0),):
        raise ValueError("corner0 must have V0 entries")
    wmax = int(V1 if wmax is None else wmax)
    i32 = C.POINTER(C.c_int32)
    n = lib().pvi_plan_plane_tiles(int(V0), int(V1), c0.ctypes.data_as(i32), int(cap), int(threads), wmax, None, 0)
    if n < 0:
        check(n)
    out = np.zeros((n, 4), dtype=np.int32)
    check(min(0, lib().pvi_plan_plane_tiles(int(V0), int(V1), c0.ctypes.data_as(i32), int(cap), int(threads), wmax,
                                           out.ctypes.data_as(i32), n)))
    return out


def plan_schedule(rows, n1, tiles_per_plane, bands):
    """Host-only diagnostic (pvi_plan_schedule): tile id of every physical block of the 4-D sweep's launch (0xffffffff = padding)."""
    n = lib().pvi_plan_schedule(int(rows), int(n1), int(tiles_per_plane), int(bands), None, 0)
    if n < 0:
        check(int(n))
    out = np.zeros(n, dtype=np.uint32)
    lib().pvi_plan_schedule(int(rows), int(n1), int(tiles_per_plane), int(bands), out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    return out


def plan_schedule_rows(n1, tiles_per_plane, bands, counts):
    """Host-only diagnostic (pvi_plan_schedule_rows): the launch order when row r of axis 0 has counts[r] tiles."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    cp = counts.ctypes.data_as(C.POINTER(C.c_int32))
    n = lib().pvi_plan_schedule_rows(len(counts), int(n1), int(tiles_per_plane), int(bands), cp, None, 0)
    if n < 0:
        check(int(n))
    out = np.zeros(n, dtype=np.uint32)
    lib().pvi_plan_schedule_rows(len(counts), int(n1), int(tiles_per_plane), int(bands), cp, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    return out


def comm_unique_id():
    """128 opaque bytes naming a new RCCL communicator (pvi_comm_unique_id): create on one rank, give to all."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    check(lib().pvi_comm_unique_id(buf))
    return bytes(buf)


class ShardedProblem:
    """One rank's slab of a grid cut along axis 0, with the halo exchange and the statistics all-reduce done by RCCL
    inside the library (pvi_shard_*).  `problem_kwargs` are the arguments of Problem for the WHOLE grid."""

    def __init__(self, rank, world, halo_rows, comm_id=None, overlap=True, transport=None, **problem_kwargs):
        self._desc_owner = Problem(_create=False, **problem_kwargs)     # keeps the host buffers alive during create
        self.rank, self.world = int(rank), int(world)
        self.plane = self._desc_owner.plane
        self.dtype = self._desc_owner.dtype
        self._h = _h()
        idp = None
        if comm_id is not None:
            if len(comm_id) != COMM_ID_BYTES:
                raise ValueError("comm_id must be %d bytes" % COMM_ID_BYTES)
            idp = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(comm_id)
        if transport is not None:
            # (sendrecv(send_lo, recv_lo, lo_send_bytes, lo_recv_bytes, send_hi, recv_hi, hi_send_bytes, hi_recv_bytes,
            #  stream) -> 0, max3(ctypes double[3]) -> 0): the caller moves the halos, e.g. over MPI or host-staged
            sr, mx = transport
            self._transport = pvi_transport(None, SENDRECV_FN(lambda u, *a: int(sr(*a) or 0)),
                                            MAX3_FN(lambda u, v: int(mx(v) or 0)))
            check(lib().pvi_shard_create_with_transport(C.byref(self._desc_owner._desc), self.rank, self.world,
                                                        int(halo_rows), C.addressof(self._transport), int(bool(overlap)),
                                                        C.byref(self._h)))
        else:
            check(lib().pvi_shard_create(C.byref(self._desc_owner._desc), self.rank, self.world, int(halo_rows), idp,
                                         int(bool(overlap)), C.byref(self._h)))
        r0, r1 = C.c_int32(), C.c_int32()
        check(lib().pvi_shard_rows(self._h, C.byref(r0), C.byref(r1)))
        self.rows = (r0.value, r1.value)
        h = C.c_int32()
        check(lib().pvi_shard_halo(self._h, C.byref(h)))
        self.halo = h.value                 # the width the ranks agreed on

    def close(self):
        if getattr(self, "_h", None):
            lib().pvi_shard_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def terminal_cost(self):
        check(lib().pvi_shard_terminal_cost(self._h))

    def sweep(self, max_sweeps, alpha=1.0, tol=-1.0):
        """-> (stats4 of the whole grid for the last sweep, sweeps done)."""
        st = np.zeros(4)
        done = C.c_int32(0)
        check(lib().pvi_shard_sweep(self._h, int(max_sweeps), float(alpha), float(tol), _ptr(st), C.byref(done)))
        return st, done.value

    def get_J(self, prev=False):
        out = np.empty((self.rows[1] - self.rows[0]) * self.plane)
        check((lib().pvi_shard_get_J_prev if prev else lib().pvi_shard_get_J)(self._h, _ptr(out)))
        return out

    def set_tables(self, x_next, G, ok=None):
        """Tier B: tables of this rank's rows ([owned nodes, A, n], [owned nodes, A]; ok mask: base-class semantics)."""
        x_next, G = _f64(x_next), _f64(G)
        nodes = (self.rows[1] - self.rows[0]) * self.plane
        A, n = self._desc_owner.actions_n, self._desc_owner.n
        if x_next.shape != (nodes, A, n) or G.shape != (nodes, A):
            raise ValueError("table shapes do not match this rank's rows")
        okp = None
        if ok is not None:
            ok = np.ascontiguousarray(ok, dtype=np.uint8)
            if ok.shape != G.shape:
                raise ValueError("ok mask shape does not match this rank's rows")
            okp = ok.ctypes.data_as(C.POINTER(C.c_uint8))
        check(lib().pvi_shard_set_tables(self._h, _ptr(x_next), _ptr(G), okp))

    def set_J(self, J):
        J = _f64(J).ravel()
        if J.size != (self.rows[1] - self.rows[0]) * self.plane:
            raise ValueError("Grid size does not match data")
        check(lib().pvi_shard_set_J(self._h, _ptr(J)))

    def get_pi(self):
        out = np.empty((self.rows[1] - self.rows[0]) * self.plane, dtype=np.int64)
        check(lib().pvi_shard_get_pi(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def describe(self):
        buf = C.create_string_buffer(2048)
        check(lib().pvi_shard_describe(self._h, buf, 1024))
        return buf.value.decode()

    def gather_J(self, prev=False):
        """J (or J_next) of the whole grid on every rank: collective over the RCCL communicator."""
        out = np.empty(self._desc_owner.dims[0] * self.plane)
        check(lib().pvi_shard_gather_J(self._h, int(bool(prev)), _ptr(out)))
        return out

    def gather_pi(self):
        out = np.empty(self._desc_owner.dims[0] * self.plane, dtype=np.int64)
        check(lib().pvi_shard_gather_pi(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def stats_every_sweep(self, on):
        check(lib().pvi_shard_stats_every_sweep(self._h, int(bool(on))))

    def sweep_history(self, max_rows=1024):
        st = np.zeros((max(int(max_rows), 1), 4))
        n = C.c_int32(0)
        check(lib().pvi_shard_sweep_history(self._h, _ptr(st), int(max_rows), C.byref(n)))
        return st[:n.value]

    def timing(self):
        """Per-sweep GPU milliseconds of the last sweep() call on this rank (pvi_shard_timing)."""
        t = np.zeros(6)
        check(lib().pvi_shard_timing(self._h, _ptr(t)))
        return dict(boundary_ms=t[0], interior_ms=t[1], exchange_ms=t[2], exposed_exchange_ms=t[3], sweep_ms=t[4],
                    sweeps_timed=int(t[5]))
