"""The C/OpenMP twin of the oracle must be bit-identical to the NumPy oracle (CPU only)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import c_oracle as CO
from oracle import vi_oracle as O
from test_oracle_golden import DOUBLEP, TWOLINK, load, problem_from

CASES = {
    "pendulum_21x21x5": (O.DYN_PENDULUM, O.pendulum_consts()),
    "pendulum_demo_51x51x9": (O.DYN_PENDULUM, O.pendulum_consts()),
    "cartpole_11p4x5": (O.DYN_CARTPOLE, O.cartpole_consts()),
    "twolink_11p4x3x3": (O.DYN_TWOLINK, O.twolink_consts(**TWOLINK)),
    "doublependulum_13x11x13x11x3x3": (O.DYN_TWOLINK, O.twolink_consts(**DOUBLEP)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_c_twin_bitwise(name):
    g = load(name)
    p = problem_from(g, *CASES[name])
    alpha = float(g["alpha"]) if "alpha" in g.files else 1.0
    c = CO.CProblem(p)
    J = O.terminal_cost(p)
    assert np.array_equal(c.terminal_cost(), J)
    for _ in range(3):
        Jn, pi = O.sweep(p, J, alpha)
        Jc, pic = c.sweep(J, alpha, threads=2)
        assert np.array_equal(Jc, Jn) and np.array_equal(pic, pi)
        J = Jn
    # partial range + f32 storage rounding
    lo, hi = p.nodes_n // 3, p.nodes_n // 2
    Jc, _ = c.sweep(J, alpha, lo, hi, f32=True)
    Jn, _ = O.sweep(p, J, alpha, ids=np.arange(lo, hi), dtype=np.float32)
    assert np.array_equal(Jc, Jn)


def test_c_twin_config1_full_solve():
    g = load("config1_pendulum_101x101x11")
    p = problem_from(g, O.DYN_PENDULUM, O.pendulum_consts())
    c = CO.CProblem(p)
    J, k, delta = c.terminal_cost(), 0, p.INF
    while delta > 0.1:
        Jn, pi = c.sweep(J)
        _, delta = O.sweep_stats(Jn, J)
        J, k = Jn, k + 1
    assert k == 618
    np.testing.assert_allclose(J, g["J"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name", ["pendulum_demo_51x51x9", "cartpole_11p4x5"])
def test_c_twin_persistent_sweeps_and_q_probe(name):
    """vio_sweeps (bench.py's persistent-region baseline) and vio_q_at (policy regret at sampled nodes) are the
    same arithmetic as vio_sweep: bit-identical to the NumPy oracle."""
    g = load(name)
    p = problem_from(g, *CASES[name])
    c = CO.CProblem(p)
    J0 = O.terminal_cost(p)
    J = J0
    for _ in range(3):
        J_prev = J
        J, pi = O.sweep(p, J)
    Jc, pic, work = c.sweeps(J0, 3, threads=2)
    assert np.array_equal(Jc, J) and np.array_equal(pic, pi)
    Jn, _ = O.sweep(p, J)
    Jn, _ = O.sweep(p, Jn)
    work[0][:] = J                                  # buffers reused: nothing is allocated between timings
    assert np.array_equal(c.sweeps(None, 2, threads=3, work=work)[0], Jn)
    rng = np.random.default_rng(0)
    nodes = rng.integers(0, p.nodes_n, 500)
    acts = rng.integers(0, p.actions_n, 500)
    q, qmin = c.q_at(J_prev, nodes, acts)
    xn, _, _, G = O.cells(p, nodes)
    Q = G + O.interp_nlinear(p.levels, np.asarray(J_prev).reshape(p.dims), xn)
    assert np.array_equal(q, Q[np.arange(500), acts]) and np.array_equal(qmin, Q.min(axis=1))
    assert np.array_equal(qmin, J[nodes])
