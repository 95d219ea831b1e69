"""
Reference-side binding of libpyrovi.so (INTEGRATION.md route B): the file a pyro maintainer would add as
pyro/planning/dynamicprogramming_hip.py.  It needs nothing from pyro_amd -- only ctypes, NumPy and the C ABI of
include/pyrovi.h -- and keeps pyro's own objects (GridDynamicSystem with its x_next_table, the cost function's G):
only the Bellman backup (pyro/planning/dynamicprogramming.py:557-570) and the statistics of
finalize_backward_step (:240-261) move to the GPU.

`HipSweepEngine` is the binding proper (plain arrays in, plain arrays out); `DynamicProgrammingHIP` is the
subclass of pyro's DynamicProgrammingWithLookUpTable and exists only when pyro itself is importable.
tests/test_gpu_parity.py::test_reference_side_stub_on_reference_tables runs the engine against tables and
results produced by the reference.
"""
import ctypes as C
import os

import numpy as np

_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)


class pvi_desc(C.Structure):            # mirrors struct pvi_desc in include/pyrovi.h (field order matters)
    _fields_ = [("struct_size", C.c_uint32), ("n", C.c_int32), ("m", C.c_int32),
                ("x_dim", C.c_int32 * 4), ("u_dim", C.c_int32 * 2),
                ("x_level", _dp * 4), ("u_level", _dp * 2),
                ("x_lb", C.c_double * 4), ("x_ub", C.c_double * 4), ("u_lb", C.c_double * 2), ("u_ub", C.c_double * 2),
                ("dt", C.c_double), ("dtype", C.c_int32), ("dynamics_id", C.c_int32),
                ("dyn_params", C.c_double * 16), ("trig", _dp * 4),
                ("cost_id", C.c_int32), ("ontarget_check", C.c_int32),
                ("Q", C.c_double * 16), ("R", C.c_double * 4), ("S", C.c_double * 16),
                ("xbar", C.c_double * 4), ("ubar", C.c_double * 2), ("EPS", C.c_double), ("INF", C.c_double),
                ("row_begin", C.c_int32), ("row_end", C.c_int32), ("halo_lo", C.c_int32), ("halo_hi", C.c_int32),
                ("device", C.c_int32), ("flags", C.c_int32), ("ext_J", C.c_void_p * 2), ("ext_pi", C.c_void_p),
                ("n_obs", C.c_int32), ("obs_axis", C.c_int32 * 2), ("obs_half", C.c_double * 2),
                ("obs_box", (C.c_double * 4) * 8), ("act_aux", _dp)]


def load_library(path=None):
    lib = C.CDLL(path or os.environ.get("LIBPYROVI", "libpyrovi.so"))
    lib.pvi_last_error.restype = C.c_char_p
    lib.pvi_create.argtypes = [C.POINTER(pvi_desc), C.POINTER(C.c_void_p)]
    lib.pvi_destroy.argtypes = [C.c_void_p]
    lib.pvi_destroy.restype = None
    lib.pvi_set_tables.argtypes = [C.c_void_p, _dp, _dp, C.POINTER(C.c_uint8)]
    lib.pvi_set_J.argtypes = [C.c_void_p, _dp, C.c_int32, C.c_int32]
    lib.pvi_get_J.argtypes = [C.c_void_p, _dp, C.c_int32, C.c_int32]
    lib.pvi_get_pi.argtypes = [C.c_void_p, _i64p, C.c_int32, C.c_int32]
    lib.pvi_sweep.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, _dp, C.POINTER(C.c_int32)]
    return lib


class HipSweepEngine:
    """The table tier of libpyrovi behind four calls: create, set_tables, set_J, sweep (+ downloads)."""

    def __init__(self, lib, x_level, u_level, x_lb, x_ub, u_lb, u_ub, dt, x_next_table, G, INF, dtype="float64"):
        self.lib = lib
        n, m = len(x_level), len(u_level)
        d = pvi_desc()
        d.struct_size = C.sizeof(pvi_desc)
        d.n, d.m = n, m
        self._keep = [np.ascontiguousarray(l, float) for l in list(x_level) + list(u_level)]
        for i in range(n):
            d.x_dim[i] = len(x_level[i])
            d.x_level[i] = self._keep[i].ctypes.data_as(_dp)
            d.x_lb[i], d.x_ub[i] = x_lb[i], x_ub[i]
        for k in range(m):
            d.u_dim[k] = len(u_level[k])
            d.u_level[k] = self._keep[n + k].ctypes.data_as(_dp)
            d.u_lb[k], d.u_ub[k] = u_lb[k], u_ub[k]
        d.dt = dt
        d.dtype = 1 if np.dtype(dtype) == np.float64 else 0       # PVI_F64 | PVI_F32
        d.dynamics_id, d.cost_id = 0, 0                            # PVI_DYN_TABLE, PVI_COST_TABLE
        d.INF = INF
        d.row_begin, d.row_end = 0, d.x_dim[0]
        self.rows = int(d.x_dim[0])
        self.nodes = int(np.prod([len(l) for l in x_level]))
        self._h = C.c_void_p()
        self._chk(lib.pvi_create(C.byref(d), C.byref(self._h)))
        xn = np.ascontiguousarray(x_next_table, float)
        Gc = np.ascontiguousarray(G, float)
        self._chk(lib.pvi_set_tables(self._h, xn.ctypes.data_as(_dp), Gc.ctypes.data_as(_dp), None))  # None: LUT semantics

    def _chk(self, rc):
        if rc:
            raise RuntimeError(self.lib.pvi_last_error().decode())

    def set_J(self, J):
        J = np.ascontiguousarray(J, float)
        self._chk(self.lib.pvi_set_J(self._h, J.ctypes.data_as(_dp), 0, self.rows))

    def sweep(self, n, alpha=1.0, tol=-1.0):
        """n backups (or until delta <= tol) without leaving the device -> (stats [done, 4], done)."""
        stats = np.zeros((max(n, 1), 4))
        done = C.c_int32()
        self._chk(self.lib.pvi_sweep(self._h, n, C.c_double(alpha), C.c_double(tol), stats.ctypes.data_as(_dp),
                                     C.byref(done)))
        return stats[:done.value], done.value

    def get_J(self):
        J = np.empty(self.nodes)
        self._chk(self.lib.pvi_get_J(self._h, J.ctypes.data_as(_dp), 0, self.rows))
        return J

    def get_pi(self):
        pi = np.empty(self.nodes, np.int64)
        self._chk(self.lib.pvi_get_pi(self._h, pi.ctypes.data_as(_i64p), 0, self.rows))
        return pi

    def close(self):
        if self._h:
            self.lib.pvi_destroy(self._h)
            self._h = C.c_void_p()


try:                                    # the subclass needs pyro itself
    from pyro.planning.dynamicprogramming import DynamicProgrammingWithLookUpTable
except ImportError:                     # (the GPU test box has no pyro: the engine above is what it exercises)
    DynamicProgrammingWithLookUpTable = None

if DynamicProgrammingWithLookUpTable is not None:

    class DynamicProgrammingHIP(DynamicProgrammingWithLookUpTable):
        """pyro's LUT value iteration with the sweeps on an MI355X (table tier: works for every pyro system)."""

        def __init__(self, grid_sys, cost_function, final_time=0, library=None):
            super().__init__(grid_sys, cost_function, final_time)        # builds grid_sys.x_next_table and self.G as usual
            g, s = grid_sys, grid_sys.sys
            self._engine = HipSweepEngine(load_library(library), g.x_level, g.u_level, s.x_lb, s.x_ub, s.u_lb, s.u_ub,
                                          g.dt, g.x_next_table, self.G, self.cf.INF)
            self._engine.set_J(self.J)

        def initialize_backward_step(self):                              # :175-193 without the host interpolant
            self.k = self.k + 1
            self.t = self.t - self.grid_sys.dt
            self.J_next = self.J

        def compute_backward_step(self):                                 # replaces :557-570
            self._stats, _ = self._engine.sweep(1, self.alpha)
            self.J = self._engine.get_J()
            self.pi = self._engine.get_pi()

        def solve_bellman_equation(self, tol=0.1, **kw):                 # :283-314 in one call, stop test on the device
            stats, n = self._engine.sweep(1 << 20, self.alpha, tol)
            self.k, self.t = self.k + n, self.t - n * self.grid_sys.dt
            self.J = self._engine.get_J()
            self.pi = self._engine.get_pi()
            return stats
