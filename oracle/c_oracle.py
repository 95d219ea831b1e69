"""ctypes front end of oracle/vi_oracle.c (test infrastructure; see that file's header)."""
import ctypes as C
import os

import numpy as np

from oracle import build_c
from oracle import vi_oracle as O

_dp = C.POINTER(C.c_double)


class vio_problem(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("A", C.c_int32), ("dyn", C.c_int32),
                ("dim", C.c_int32 * 4), ("lev", _dp * 4), ("trig", _dp * 4), ("utab", _dp),
                ("lb", C.c_double * 4), ("ub", C.c_double * 4), ("u_lb", C.c_double * 2), ("u_ub", C.c_double * 2),
                ("dt", C.c_double), ("c", C.c_double * 16), ("Q", C.c_double * 16), ("R", C.c_double * 4),
                ("S", C.c_double * 16), ("xbar", C.c_double * 4), ("ubar", C.c_double * 2),
                ("EPS", C.c_double), ("INF", C.c_double), ("ontarget", C.c_int32)]


_lib = None
# CPUs this process may run on, taken BEFORE libgomp is loaded: with OMP_PROC_BIND set the runtime pins the calling
# thread to one place, after which sched_getaffinity(0) reports a single CPU
try:
    ALLOWED_CPUS = frozenset(os.sched_getaffinity(0))
except AttributeError:
    ALLOWED_CPUS = frozenset(range(os.cpu_count() or 1))


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_c.build())
        L.vio_sweep.restype = None
        L.vio_sweep.argtypes = [C.POINTER(vio_problem), _dp, _dp, C.POINTER(C.c_int64), C.c_double, C.c_int64,
                                C.c_int64, C.c_int32, C.c_int32]
        L.vio_sweeps.restype = None
        L.vio_sweeps.argtypes = [C.POINTER(vio_problem), _dp, _dp, C.POINTER(C.c_int64), C.c_double, C.c_int32,
                                 C.c_int32, C.c_int32]
        L.vio_q_at.restype = None
        L.vio_q_at.argtypes = [C.POINTER(vio_problem), _dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int64,
                               C.c_double, _dp, _dp]
        L.vio_terminal_cost.restype = None
        L.vio_terminal_cost.argtypes = [C.POINTER(vio_problem), _dp]
        L.vio_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(_dp)


class CProblem:
    """Wraps an oracle Problem (vi_oracle.Problem) for the C twin."""

    def __init__(self, p: O.Problem):
        self.p = p
        self._keep = []
        d = vio_problem()
        d.n, d.m, d.A, d.dyn = p.n, p.m, p.actions_n, p.dyn_id
        for i, l in enumerate(p.levels):
            a = np.ascontiguousarray(l, dtype=np.float64); self._keep.append(a)
            d.dim[i], d.lev[i] = len(a), _ptr(a)
            d.lb[i], d.ub[i] = p.x_lb[i], p.x_ub[i]
            d.xbar[i] = p.xbar[i]
        t = p.trig_tables()
        order = {O.DYN_PENDULUM: ["s0"], O.DYN_CARTPOLE: ["c1", "s1"], O.DYN_TWOLINK: ["s0", "c1", "s1", "s01"]}
        for i, k in enumerate(order[p.dyn_id]):
            a = np.ascontiguousarray(t[k], dtype=np.float64); self._keep.append(a)
            d.trig[i] = _ptr(a)
        ut = np.ascontiguousarray(p.u_table, dtype=np.float64); self._keep.append(ut)
        d.utab = _ptr(ut)
        for k in range(p.m):
            d.u_lb[k], d.u_ub[k], d.ubar[k] = p.u_lb[k], p.u_ub[k], p.ubar[k]
        d.dt = p.dt
        d.c[:len(p.dyn_c)] = list(p.dyn_c)
        d.Q[:p.n * p.n] = list(np.asarray(p.Q, dtype=float).ravel())
        d.S[:p.n * p.n] = list(np.asarray(p.S, dtype=float).ravel())
        d.R[:p.m * p.m] = list(np.asarray(p.R, dtype=float).ravel())
        d.EPS, d.INF, d.ontarget = p.EPS, p.INF, int(p.ontarget_check)
        self.d = d

    def terminal_cost(self):
        J = np.empty(self.p.nodes_n)
        lib().vio_terminal_cost(C.byref(self.d), _ptr(J))
        return J

    def sweep(self, J, alpha=1.0, node0=0, node1=None, f32=False, threads=0):
        """Backup of nodes [node0,node1) -> (J_new[node1-node0], pi)."""
        node1 = self.p.nodes_n if node1 is None else node1
        Jin = np.ascontiguousarray(J, dtype=np.float64)
        out = np.empty(self.p.nodes_n)
        pi = np.empty(self.p.nodes_n, dtype=np.int64)
        lib().vio_sweep(C.byref(self.d), _ptr(Jin), _ptr(out), pi.ctypes.data_as(C.POINTER(C.c_int64)), float(alpha),
                        int(node0), int(node1), int(f32), int(threads))
        return out[node0:node1], pi[node0:node1]


    def sweeps(self, J0, nsweeps, alpha=1.0, f32=False, threads=0, work=None):
        """`nsweeps` whole-grid backups in one persistent parallel region (bench.py's cpu_baseline).  `work` =
        (J0, J1, pi) buffers from a previous call are reused, so repeated timings allocate nothing."""
        if work is None:
            work = (np.array(J0, dtype=np.float64), np.zeros(self.p.nodes_n), np.zeros(self.p.nodes_n, dtype=np.int64))
        a, b, pi = work
        lib().vio_sweeps(C.byref(self.d), _ptr(a), _ptr(b), pi.ctypes.data_as(C.POINTER(C.c_int64)), float(alpha),
                         int(nsweeps), int(f32), int(threads))
        return (b if nsweeps & 1 else a), pi, work

    def q_at(self, J, nodes, actions, alpha=1.0):
        """(Q[s, a_s], min_a Q[s, a]) for the given node ids and one action each."""
        Jin = np.ascontiguousarray(J, dtype=np.float64)
        nodes = np.ascontiguousarray(nodes, dtype=np.int64)
        actions = np.ascontiguousarray(actions, dtype=np.int64)
        q, qmin = np.empty(len(nodes)), np.empty(len(nodes))
        i64 = C.POINTER(C.c_int64)
        lib().vio_q_at(C.byref(self.d), _ptr(Jin), nodes.ctypes.data_as(i64), actions.ctypes.data_as(i64), len(nodes),
                       float(alpha), _ptr(q), _ptr(qmin))
        return q, qmin


def max_threads():
    return lib().vio_max_threads()


def physical_cores():
    """Distinct (package, core) pairs this process may run on: SMT siblings share one set of execution units, so the
    all-cores baseline uses one thread per physical core."""
    seen = set()
    for cpu in ALLOWED_CPUS:
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % cpu
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            seen.add(("?", cpu))
    return max(1, len(seen))
