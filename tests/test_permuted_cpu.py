"""
CPU tests of the internal axis order (pyro_amd/planning/permuted.py, DynamicProgramming(internal_order="swapped")): the swapped
problem is the SAME problem -- checked on the oracle, whose table-driven sweep (oracle/vi_oracle.py sweep_lut, the reference's
dynamicprogramming.py:564-570) runs the cart-pole with its coordinates swapped from tables built by a host restatement of that
system, against the oracle's closed form in the reference's order -- and the wrapper hands node-ordered arrays over transposed.
The GPU side (Dyn<PVI_DYN_CARTPOLE_SW>, its own dynamics id) is tests/test_gpu_zz_unproven.py::test_swapped_internal_order_*.
"""
import contextlib
import io

import numpy as np
import pytest

from oracle import vi_oracle as O
from pyro_amd import _native
from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import cartpole, mechanical
from pyro_amd.planning import discretizer, permuted


class _CartPoleThetaFirst(mechanical.MechanicalSystem):
    """reference pyro/dynamic/cartpole.py:322-437 with the generalised coordinates in the order q = (theta, x)"""

    def __init__(self):
        super().__init__(dof=2, actuators=1)
        self.u_lb[0], self.u_ub[0] = -10, +10
        self.lcg, self.m1, self.m2, self.gravity = 0.5, 1, 0.1, 9.81

    def H(self, q):
        off = self.m2 * self.lcg * np.cos(q[0])
        return np.array([[self.m2 * self.lcg ** 2, off], [off, self.m1 + self.m2]], dtype=float)

    def C(self, q, dq):
        C = np.zeros((2, 2))
        C[1, 0] = -self.m2 * self.lcg * np.sin(q[0]) * dq[0]
        return C

    def B(self, q):
        return np.array([[0.0], [1.0]])

    def g(self, q):
        return np.array([self.m2 * self.gravity * self.lcg * np.sin(q[0]), 0.0])

    def d(self, q, dq):
        return np.zeros(2)


def _grid(dims=(7, 9, 5, 6), nact=3):
    s = cartpole.CartPole()
    s.x_lb, s.x_ub = np.array([-2.0, -3.0, -4.0, -5.0]), np.array([2.5, 3.5, 4.5, 5.5])     # every axis its own box
    s.xbar = np.array([0.3, np.pi - 3.0, -0.2, 0.4])
    with contextlib.redirect_stdout(io.StringIO()):
        g = discretizer.GridDynamicSystem(s, list(dims), [nact], 0.05, lookup=False)
    cf = costfunction.QuadraticCostFunction.from_sys(s)
    cf.Q = np.diag([1.0, 2.0, 3.0, 4.0]) + 0.1 * (np.ones((4, 4)) - np.eye(4))                # symmetric, not diagonal
    cf.S = np.diag([0.5, 0.0, 0.25, 0.125])
    cf.INF = 1000
    return s, g, cf


def test_swapped_problem_kwargs_describe_the_same_problem():
    """J of the swapped problem -- its tables from a host restatement of the cart-pole with q = (theta, x), its levels / box /
    cost from swap_problem_kwargs -- is the reference-order J with the axes (0 1)(2 3) exchanged, sweep after sweep (float64:
    only the order of the corner sum differs)."""
    s, g, cf = _grid()
    kw = g._problem_kwargs(cf.device_cost(), "float32")
    assert kw["dynamics_id"] == _native.DYN_CARTPOLE
    sk = permuted.swap_problem_kwargs(kw)
    assert [len(l) for l in sk["x_levels"]] == [9, 7, 6, 5] and sk["dynamics_id"] == _native.DYN_CARTPOLE_SW and list(sk["dyn_params"]) == list(kw["dyn_params"])
    assert np.array_equal(sk["x_lb"], s.x_lb[[1, 0, 3, 2]]) and np.array_equal(sk["cost"]["xbar"], cf.xbar[[1, 0, 3, 2]])
    assert sk["cost"]["Q"][0, 0] == 2.0 and sk["cost"]["Q"][3, 3] == 3.0 and sk["cost"]["S"][2, 2] == 0.125
    assert kw["dynamics_id"] == _native.DYN_CARTPOLE                         # the caller's arguments are untouched
    # reference order: the oracle's closed form
    dyn_id, params = s.device_dynamics()
    p = O.Problem(g.x_level, g.u_level, g.dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar, float(cf.INF), float(cf.EPS),
                  x_lb=s.x_lb, x_ub=s.x_ub)
    J = O.terminal_cost(p)
    # swapped order: tables from the theta-first system, cost and box from the swapped arguments
    t = _CartPoleThetaFirst()
    lv, c = sk["x_levels"], sk["cost"]
    dims = tuple(len(l) for l in lv)
    X = np.stack([a.ravel() for a in np.meshgrid(*lv, indexing="ij")], axis=-1)
    U = np.asarray(g.u_level[0])
    xn = np.empty((len(X), len(U), 4))
    G = np.empty((len(X), len(U)))
    for i, x in enumerate(X):
        dx = x - c["xbar"]
        gx = dx @ c["Q"] @ dx
        on = np.linalg.norm(dx) < c["EPS"]
        for a, u in enumerate(U):
            xn[i, a] = t.f(x, np.array([u])) * g.dt + x
            ok = np.all(xn[i, a] >= sk["x_lb"]) and np.all(xn[i, a] <= sk["x_ub"])
            G[i, a] = (0.0 if on else (gx + u * cf.R[0, 0] * u) * g.dt) if ok else c["INF"]
    dxs = X - c["xbar"]
    Js = np.einsum("ni,ij,nj->n", dxs, c["S"], dxs)
    Js[np.linalg.norm(dxs, axis=1) < c["EPS"]] = 0.0
    for k in range(6):
        assert np.allclose(Js.reshape(dims).transpose(1, 0, 3, 2).ravel(), J, rtol=1e-12, atol=1e-12), k
        J, pi = O.sweep(p, J)
        Js, pis, _ = O.sweep_lut(lv, xn, G, Js)
    assert np.abs(J).max() > 1.0 and len(np.unique(pi)) > 1
    assert np.array_equal(pis.reshape(dims).transpose(1, 0, 3, 2).ravel(), pi)


class _FakeInner:
    def __init__(self, dims):
        self.dims, self.J, self.pi, self.calls = tuple(dims), None, None, []

    def set_J(self, J):
        self.J = np.array(J, dtype=float)

    def get_J(self, prev=False):
        self.calls.append(("get_J", prev))
        return self.J.copy()

    def set_pi(self, pi):
        self.pi = np.array(pi)

    def get_pi(self):
        return self.pi.copy()

    def describe(self):
        return "path=fake"

    def sweep(self, n, alpha=1.0, tol=-1.0):
        return [(0.0, 0.0, 0.0, 0.0)] * n, n


def test_swapped_problem_hands_node_ordered_arrays_over_transposed():
    dims = (3, 4, 5, 6)
    inner = _FakeInner((4, 3, 6, 5))
    p = permuted.SwappedProblem(inner, dims)
    J = np.arange(np.prod(dims), dtype=float)
    p.set_J(J)
    # inside: node (i1, i0, i3, i2) of the swapped grid holds the reference's node (i0, i1, i2, i3)
    assert inner.J.reshape(4, 3, 6, 5)[2, 1, 4, 3] == J.reshape(dims)[1, 2, 3, 4]
    assert np.array_equal(p.get_J(), J) and np.array_equal(p.get_J(prev=True), J) and inner.calls[-1] == ("get_J", True)
    pi = (np.arange(np.prod(dims)) % 7).astype(np.int64)
    p.set_pi(pi)
    assert np.array_equal(p.get_pi(), pi) and inner.pi.reshape(4, 3, 6, 5)[0, 2, 1, 4] == pi.reshape(dims)[2, 0, 4, 1]
    assert p.sweep(2)[1] == 2 and p.describe().endswith("order=swapped") and p.swapped
    with pytest.raises(ValueError):
        p.set_J(J[:-1])
    for call in (lambda: p.rollout(None, 1, 0.1), lambda: p.build_tables(), lambda: p.get_J(0, 1), lambda: p.set_interpolation("nearest")):
        with pytest.raises(NotImplementedError):
            call()
    with pytest.raises(ValueError):
        permuted.SwappedProblem(_FakeInner(dims), dims)          # the inner handle must be the swapped grid


def test_swapped_order_is_refused_where_it_is_not_defined():
    from pyro_amd.dynamic import pendulum
    s, g, cf = _grid()
    kw = g._problem_kwargs(cf.device_cost(), "float32")
    assert permuted.swap_applies(s, s.device_dynamics(), "float32") and not permuted.swap_applies(s, s.device_dynamics(), "float64")
    ps = pendulum.SinglePendulum()
    assert not permuted.swap_applies(ps, ps.device_dynamics(), "float32")
    for bad in (dict(kw, dynamics_id=_native.DYN_TWOLINK), dict(kw, rows=(0, 3))):
        with pytest.raises(NotImplementedError):
            permuted.swap_problem_kwargs(bad)
